// flux_net.cpp -- see flux_net.hpp.
#include <algorithm>
#include "flux_net.hpp"

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>

namespace flux_host {

using cbor::Decoder;
using cbor::Encoder;
using cbor::Type;

// =================================================================================================
// encoding (serde derive layouts; enums in serde_cbor <= 0.9's array form, cbor.hpp)
// =================================================================================================
namespace {

void enc_vec3(Encoder &e, const Vec3 &v) {  // nalgebra Point3 / Vector3: a 3-sequence
    e.array(3);
    e.real(v.x);
    e.real(v.y);
    e.real(v.z);
}

void enc_color(Encoder &e, const Color &c) {  // color.rs:12-16
    e.map(3);
    e.key("r");
    e.real(c.r);
    e.key("g");
    e.real(c.g);
    e.key("b");
    e.real(c.b);
}

void enc_material(Encoder &e, const MaterialData &m) {  // shapes.rs:42-81
    e.array(2);
    if (auto *a = std::get_if<MatteData>(&m)) {
        e.text("Matte");
        e.map(3);
        e.key("diffuse_color");
        enc_color(e, a->diffuse_color);
        e.key("ambient_color");
        enc_color(e, a->ambient_color);
        e.key("diffuse_coefficient");
        e.real(a->diffuse_coefficient);
    } else if (auto *b = std::get_if<EmissiveData>(&m)) {
        e.text("Emissive");
        e.map(2);
        e.key("color");
        enc_color(e, b->color);
        e.key("power");
        e.real(b->power);
    } else if (auto *c = std::get_if<ReflectiveData>(&m)) {
        e.text("Reflective");
        e.map(2);
        e.key("reflect_amount");
        e.real(c->reflect_amount);
        e.key("reflect_color");
        enc_color(e, c->reflect_color);
    } else {
        const auto &g = std::get<GlossyReflectiveData>(m);
        e.text("GlossyReflective");
        e.map(3);
        e.key("reflect_amount");
        e.real(g.reflect_amount);
        e.key("reflect_color");
        enc_color(e, g.reflect_color);
        e.key("reflect_exponent");
        e.real(g.reflect_exponent);
    }
}

void enc_shape(Encoder &e, const ShapeData &s) {  // scene.rs:71-74, shapes.rs:15-40
    e.array(2);
    if (auto *sp = std::get_if<SphereData>(&s)) {
        e.text("Sphere");
        e.map(4);
        e.key("center");
        enc_vec3(e, sp->center);
        e.key("radius");
        e.real(sp->radius);
        e.key("material");
        enc_material(e, sp->material);
        e.key("invert");
        e.boolean(sp->invert);
    } else {
        const auto &pl = std::get<PlaneData>(s);
        e.text("Plane");
        e.map(3);
        e.key("point");
        enc_vec3(e, pl.point);
        e.key("normal");
        enc_vec3(e, pl.normal);
        e.key("material");
        enc_material(e, pl.material);
    }
}

void enc_job_id(Encoder &e, const JobID &id) {  // tuple struct JobID(usize, usize), job.rs:12
    e.array(2);
    e.uint(id.allocator_id);
    e.uint(id.id);
}

void enc_scene(Encoder &e, const SceneData &s) {  // scene.rs:40-66
    e.map(6);
    e.key("scene_name");
    e.text(s.scene_name);
    e.key("output_settings");
    e.map(3);
    e.key("image_width");
    e.uint(s.output_settings.image_width);
    e.key("image_height");
    e.uint(s.output_settings.image_height);
    e.key("pixel_size");
    e.real(s.output_settings.pixel_size);
    e.key("background");
    enc_color(e, s.background);
    e.key("shapes");
    e.array(s.shapes.size());
    for (const auto &sh : s.shapes) enc_shape(e, sh);
    e.key("camera_settings");
    e.map(3);
    e.key("eye");
    enc_vec3(e, s.camera_settings.eye);
    e.key("look_at");
    enc_vec3(e, s.camera_settings.look_at);
    e.key("up");
    enc_vec3(e, s.camera_settings.up);
    e.key("camera_data");
    e.map(4);
    e.key("zoom_factor");
    e.real(s.camera_data.zoom_factor);
    e.key("view_plane_distance");
    e.real(s.camera_data.view_plane_distance);
    e.key("focal_distance");
    e.real(s.camera_data.focal_distance);
    e.key("lens_radius");
    e.real(s.camera_data.lens_radius);
}

void enc_job(Encoder &e, const Job &j) {  // job.rs:59-63
    e.map(3);
    e.key("id");
    enc_job_id(e, j.id);
    e.key("scene_data");
    enc_scene(e, j.scene_data);
    e.key("config");
    e.map(3);
    e.key("sample_root");
    e.uint(j.config.sample_root);
    e.key("max_trace_depth");
    e.uint(j.config.max_trace_depth);
    e.key("rows_per_work_unit");
    e.uint(j.config.rows_per_work_unit);
}

void enc_unit(Encoder &e, const WorkUnit &u) {  // job.rs:40-44
    e.map(3);
    e.key("row_start");
    e.uint(u.row_start);
    e.key("row_end");
    e.uint(u.row_end);
    e.key("job_id");
    enc_job_id(e, u.job_id);
}

void enc_time(Encoder &e, double t) {  // serde's SystemTime: {secs_since_epoch, nanos_since_epoch}
    double secs = std::floor(t);
    uint64_t nanos = (uint64_t)((t - secs) * 1e9);
    if (nanos > 999999999ull) nanos = 999999999ull;
    e.map(2);
    e.key("secs_since_epoch");
    e.uint(secs < 0 ? 0 : (uint64_t)secs);
    e.key("nanos_since_epoch");
    e.uint(nanos);
}

}  // namespace

void encode_worker_info(Encoder &e, const WorkerInfo &w) {  // manager.rs:221-224
    e.map(1);
    e.key("num_threads");
    e.uint(w.num_threads);
}

void encode_request(Encoder &e, const NetworkWorkerRequest &r) {
    switch (r.kind) {
        case NetworkWorkerRequest::SetJob:
            e.array(2);
            e.text("SetJob");
            enc_job(e, r.job);
            break;
        case NetworkWorkerRequest::WorkUnitMsg:
            e.array(2);
            e.text("WorkUnit");
            enc_unit(e, r.unit);
            break;
        case NetworkWorkerRequest::Done:
            e.text("Done");
            break;
    }
}

void encode_event(Encoder &e, const RenderEvent &ev) {  // manager.rs:16-28
    switch (ev.kind) {
        case RenderEvent::RenderingStarted:
            e.array(2);
            e.text("RenderingStarted");
            e.map(2);
            e.key("job_id");
            enc_job_id(e, ev.job_id);
            e.key("start_time");
            enc_time(e, ev.time_s);
            break;
        case RenderEvent::ImageInfo:
            e.array(2);
            e.text("ImageInfo");
            e.map(3);
            e.key("scene_name");
            e.text(ev.scene_name);
            e.key("width");
            e.uint(ev.width);
            e.key("height");
            e.uint(ev.height);
            break;
        case RenderEvent::RowsReady:
            e.array(2);
            e.text("RowsReady");
            e.map(2);
            e.key("work_unit");
            enc_unit(e, ev.result.work_unit);
            e.key("rows");
            e.array(ev.result.rows.size());
            for (const auto &row : ev.result.rows) {
                e.array(row.size());
                for (const Color &c : row) enc_color(e, c);
            }
            break;
        case RenderEvent::RenderingFinished:
            e.array(2);
            e.text("RenderingFinished");
            e.map(1);
            e.key("end_time");
            enc_time(e, ev.time_s);
            break;
    }
}

// =================================================================================================
// decoding
// =================================================================================================
namespace {

// map with text keys -> field(key) reads the value; unknown keys are skipped (serde ignores them)
bool read_struct(Decoder &d, const std::function<bool(const std::string &)> &field) {
    uint64_t n;
    if (!d.read_map(n)) return false;
    std::string key;
    for (uint64_t k = 0; n == Decoder::kIndefinite ? !d.at_break() : k < n; k++) {
        if (d.failed() || !d.read_text(key)) return false;
        if (!field(key)) return false;
    }
    return !d.failed();
}

// enum in any of the three layouts; payload(name) reads the variant's single payload value (not called for
// unit variants written as a bare string)
bool read_enum(Decoder &d, const std::function<bool(const std::string &, bool has_payload)> &variant) {
    std::string name;
    const Type t = d.peek();
    if (t == Type::Text) {
        return d.read_text(name) && variant(name, false);
    }
    if (t == Type::Array) {  // serde_cbor <= 0.9: [name, payload]
        uint64_t n;
        if (!d.read_array(n) || !d.read_text(name)) return false;
        if (n == 1) return variant(name, false);
        if (!variant(name, true)) return false;
        if (n == Decoder::kIndefinite) {
            while (!d.at_break())
                if (d.failed() || !d.skip()) return false;
        } else {
            for (uint64_t k = 2; k < n; k++)
                if (!d.skip()) return false;
        }
        return !d.failed();
    }
    if (t == Type::Map) {  // serde_cbor >= 0.10: {name: payload}
        uint64_t n;
        if (!d.read_map(n) || !d.read_text(name) || !variant(name, true)) return false;
        if (n == Decoder::kIndefinite) return d.at_break();
        return n == 1;
    }
    return false;
}

bool dec_usize(Decoder &d, size_t &v) {
    uint64_t u;
    if (!d.read_uint(u)) return false;
    v = (size_t)u;
    return true;
}

// 3 numbers as a sequence (nalgebra) -- or a map with x/y/z, or r/g/b for colours written as sequences
bool dec_triple(Decoder &d, double &a, double &b, double &c, const char *ka, const char *kb, const char *kc) {
    if (d.peek() == Type::Array) {
        uint64_t n;
        if (!d.read_array(n)) return false;
        if (!d.read_number(a) || !d.read_number(b) || !d.read_number(c)) return false;
        if (n == Decoder::kIndefinite) return d.at_break();
        return n == 3;
    }
    return read_struct(d, [&](const std::string &k) {
        if (k == ka) return d.read_number(a);
        if (k == kb) return d.read_number(b);
        if (k == kc) return d.read_number(c);
        return d.skip();
    });
}
bool dec_vec3(Decoder &d, Vec3 &v) { return dec_triple(d, v.x, v.y, v.z, "x", "y", "z"); }
bool dec_color(Decoder &d, Color &c) { return dec_triple(d, c.r, c.g, c.b, "r", "g", "b"); }

bool dec_material(Decoder &d, MaterialData &m) {
    return read_enum(d, [&](const std::string &name, bool has) {
        if (!has) return false;
        if (name == "Matte") {
            MatteData v;
            if (!read_struct(d, [&](const std::string &k) {
                    if (k == "diffuse_color") return dec_color(d, v.diffuse_color);
                    if (k == "ambient_color") return dec_color(d, v.ambient_color);
                    if (k == "diffuse_coefficient") return d.read_number(v.diffuse_coefficient);
                    return d.skip();
                }))
                return false;
            m = v;
            return true;
        }
        if (name == "Emissive") {
            EmissiveData v;
            if (!read_struct(d, [&](const std::string &k) {
                    if (k == "color") return dec_color(d, v.color);
                    if (k == "power") return d.read_number(v.power);
                    return d.skip();
                }))
                return false;
            m = v;
            return true;
        }
        if (name == "Reflective") {
            ReflectiveData v;
            if (!read_struct(d, [&](const std::string &k) {
                    if (k == "reflect_amount") return d.read_number(v.reflect_amount);
                    if (k == "reflect_color") return dec_color(d, v.reflect_color);
                    return d.skip();
                }))
                return false;
            m = v;
            return true;
        }
        if (name == "GlossyReflective") {
            GlossyReflectiveData v;
            if (!read_struct(d, [&](const std::string &k) {
                    if (k == "reflect_amount") return d.read_number(v.reflect_amount);
                    if (k == "reflect_color") return dec_color(d, v.reflect_color);
                    if (k == "reflect_exponent") return d.read_number(v.reflect_exponent);
                    return d.skip();
                }))
                return false;
            m = v;
            return true;
        }
        return false;  // unknown variant
    });
}

bool dec_shape(Decoder &d, ShapeData &s) {
    return read_enum(d, [&](const std::string &name, bool has) {
        if (!has) return false;
        if (name == "Sphere") {
            SphereData v;
            if (!read_struct(d, [&](const std::string &k) {
                    if (k == "center") return dec_vec3(d, v.center);
                    if (k == "radius") return d.read_number(v.radius);
                    if (k == "material") return dec_material(d, v.material);
                    if (k == "invert") return d.read_bool(v.invert);
                    return d.skip();
                }))
                return false;
            s = v;
            return true;
        }
        if (name == "Plane") {
            PlaneData v;
            if (!read_struct(d, [&](const std::string &k) {
                    if (k == "point") return dec_vec3(d, v.point);
                    if (k == "normal") return dec_vec3(d, v.normal);
                    if (k == "material") return dec_material(d, v.material);
                    return d.skip();
                }))
                return false;
            s = v;
            return true;
        }
        return false;
    });
}

bool dec_job_id(Decoder &d, JobID &id) {
    uint64_t n;
    if (!d.read_array(n) || !dec_usize(d, id.allocator_id) || !dec_usize(d, id.id)) return false;
    if (n == Decoder::kIndefinite) return d.at_break();
    return n == 2;
}

bool dec_scene(Decoder &d, SceneData &s) {
    return read_struct(d, [&](const std::string &k) {
        if (k == "scene_name") return d.read_text(s.scene_name);
        if (k == "output_settings")
            return read_struct(d, [&](const std::string &f) {
                if (f == "image_width") return dec_usize(d, s.output_settings.image_width);
                if (f == "image_height") return dec_usize(d, s.output_settings.image_height);
                if (f == "pixel_size") return d.read_number(s.output_settings.pixel_size);
                return d.skip();
            });
        if (k == "background") return dec_color(d, s.background);
        if (k == "shapes") {
            uint64_t n;
            if (!d.read_array(n)) return false;
            s.shapes.clear();
            for (uint64_t i = 0; n == Decoder::kIndefinite ? !d.at_break() : i < n; i++) {
                ShapeData sh;
                if (d.failed() || !dec_shape(d, sh)) return false;
                s.shapes.push_back(sh);
            }
            return !d.failed();
        }
        if (k == "camera_settings")
            return read_struct(d, [&](const std::string &f) {
                if (f == "eye") return dec_vec3(d, s.camera_settings.eye);
                if (f == "look_at") return dec_vec3(d, s.camera_settings.look_at);
                if (f == "up") return dec_vec3(d, s.camera_settings.up);
                return d.skip();
            });
        if (k == "camera_data")
            return read_struct(d, [&](const std::string &f) {
                if (f == "zoom_factor") return d.read_number(s.camera_data.zoom_factor);
                if (f == "view_plane_distance") return d.read_number(s.camera_data.view_plane_distance);
                if (f == "focal_distance") return d.read_number(s.camera_data.focal_distance);
                if (f == "lens_radius") return d.read_number(s.camera_data.lens_radius);
                return d.skip();
            });
        return d.skip();
    });
}

bool dec_job(Decoder &d, Job &j) {
    return read_struct(d, [&](const std::string &k) {
        if (k == "id") return dec_job_id(d, j.id);
        if (k == "scene_data") return dec_scene(d, j.scene_data);
        if (k == "config")
            return read_struct(d, [&](const std::string &f) {
                if (f == "sample_root") return dec_usize(d, j.config.sample_root);
                if (f == "max_trace_depth") return dec_usize(d, j.config.max_trace_depth);
                if (f == "rows_per_work_unit") return dec_usize(d, j.config.rows_per_work_unit);
                return d.skip();
            });
        return d.skip();
    });
}

bool dec_unit(Decoder &d, WorkUnit &u) {
    return read_struct(d, [&](const std::string &k) {
        if (k == "row_start") return dec_usize(d, u.row_start);
        if (k == "row_end") return dec_usize(d, u.row_end);
        if (k == "job_id") return dec_job_id(d, u.job_id);
        return d.skip();
    });
}

bool dec_time(Decoder &d, double &t) {
    uint64_t secs = 0, nanos = 0;
    if (!read_struct(d, [&](const std::string &k) {
            if (k == "secs_since_epoch") return d.read_uint(secs);
            if (k == "nanos_since_epoch") return d.read_uint(nanos);
            return d.skip();
        }))
        return false;
    t = (double)secs + (double)nanos * 1e-9;
    return true;
}

}  // namespace

bool decode_worker_info(Decoder &d, WorkerInfo &w) {
    return read_struct(d, [&](const std::string &k) {
        if (k == "num_threads") return dec_usize(d, w.num_threads);
        return d.skip();
    });
}

bool decode_request(Decoder &d, NetworkWorkerRequest &r) {
    return read_enum(d, [&](const std::string &name, bool has) {
        if (name == "Done" && !has) {
            r.kind = NetworkWorkerRequest::Done;
            return true;
        }
        if (name == "SetJob" && has) {
            r.kind = NetworkWorkerRequest::SetJob;
            return dec_job(d, r.job);
        }
        if (name == "WorkUnit" && has) {
            r.kind = NetworkWorkerRequest::WorkUnitMsg;
            return dec_unit(d, r.unit);
        }
        return false;
    });
}

bool decode_event(Decoder &d, RenderEvent &ev) {
    return read_enum(d, [&](const std::string &name, bool has) {
        if (!has) return false;
        if (name == "RowsReady") {
            ev.kind = RenderEvent::RowsReady;
            return read_struct(d, [&](const std::string &k) {
                if (k == "work_unit") return dec_unit(d, ev.result.work_unit);
                if (k == "rows") {
                    uint64_t n;
                    if (!d.read_array(n)) return false;
                    ev.result.rows.clear();
                    for (uint64_t i = 0; n == Decoder::kIndefinite ? !d.at_break() : i < n; i++) {
                        uint64_t m;
                        if (d.failed() || !d.read_array(m)) return false;
                        std::vector<Color> row;
                        if (m != Decoder::kIndefinite) row.reserve((size_t)std::min<uint64_t>(m, 1u << 16));  // the count is peer-controlled
                        for (uint64_t c = 0; m == Decoder::kIndefinite ? !d.at_break() : c < m; c++) {
                            Color col;
                            if (d.failed() || !dec_color(d, col)) return false;
                            row.push_back(col);
                        }
                        ev.result.rows.push_back(std::move(row));
                    }
                    return !d.failed();
                }
                return d.skip();
            });
        }
        if (name == "ImageInfo") {
            ev.kind = RenderEvent::ImageInfo;
            return read_struct(d, [&](const std::string &k) {
                if (k == "scene_name") return d.read_text(ev.scene_name);
                if (k == "width") return dec_usize(d, ev.width);
                if (k == "height") return dec_usize(d, ev.height);
                return d.skip();
            });
        }
        if (name == "RenderingStarted") {
            ev.kind = RenderEvent::RenderingStarted;
            return read_struct(d, [&](const std::string &k) {
                if (k == "job_id") return dec_job_id(d, ev.job_id);
                if (k == "start_time") return dec_time(d, ev.time_s);
                return d.skip();
            });
        }
        if (name == "RenderingFinished") {
            ev.kind = RenderEvent::RenderingFinished;
            return read_struct(d, [&](const std::string &k) {
                if (k == "end_time") return dec_time(d, ev.time_s);
                return d.skip();
            });
        }
        return false;
    });
}

// =================================================================================================
// TCP
// =================================================================================================
TcpStream::~TcpStream() {
    if (fd_ >= 0) ::close(fd_);
}

bool TcpStream::read(void *dst, size_t n) {
    char *p = static_cast<char *>(dst);
    while (n > 0) {
        const ssize_t got = ::recv(fd_, p, n, 0);
        if (got > 0) {
            p += got;
            n -= (size_t)got;
        } else if (got < 0 && errno == EINTR) {
            continue;
        } else {
            return false;
        }
    }
    return true;
}

bool TcpStream::write_all(const std::string &bytes) {
    const char *p = bytes.data();
    size_t n = bytes.size();
    while (n > 0) {
        const ssize_t put = ::send(fd_, p, n, MSG_NOSIGNAL);
        if (put > 0) {
            p += put;
            n -= (size_t)put;
        } else if (put < 0 && errno == EINTR) {
            continue;
        } else {
            return false;
        }
    }
    return true;
}

void TcpStream::shutdown_both() {
    if (fd_ >= 0) ::shutdown(fd_, SHUT_RDWR);
}

std::string TcpStream::peer() const {
    sockaddr_storage ss;
    socklen_t len = sizeof ss;
    char host[NI_MAXHOST] = "?", serv[NI_MAXSERV] = "?";
    if (::getpeername(fd_, (sockaddr *)&ss, &len) == 0)
        ::getnameinfo((sockaddr *)&ss, len, host, sizeof host, serv, sizeof serv, NI_NUMERICHOST | NI_NUMERICSERV);
    return std::string(host) + ":" + serv;
}

static void split_endpoint(const std::string &raw, std::string &host, std::string &port) {
    const size_t colon = raw.find(':');  // workers.rs:120-123: no ':' -> default port
    if (colon == std::string::npos) {
        host = raw;
        port = kDefaultPort;
    } else {
        host = raw.substr(0, colon);
        port = raw.substr(colon + 1);
    }
}

std::unique_ptr<TcpStream> TcpStream::connect(const std::string &endpoint) {
    std::string host, port;
    split_endpoint(endpoint, host, port);
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_UNSPEC;
    hints.ai_socktype = SOCK_STREAM;
    const int rc = ::getaddrinfo(host.c_str(), port.c_str(), &hints, &res);
    if (rc != 0) throw FluxError(FLUX_E_IO, "cannot resolve " + endpoint + ": " + gai_strerror(rc));
    int fd = -1;
    std::string last = "no address";
    for (addrinfo *a = res; a; a = a->ai_next) {
        fd = ::socket(a->ai_family, a->ai_socktype, a->ai_protocol);
        if (fd < 0) continue;
        if (::connect(fd, a->ai_addr, a->ai_addrlen) == 0) break;
        last = std::strerror(errno);
        ::close(fd);
        fd = -1;
    }
    ::freeaddrinfo(res);
    if (fd < 0) throw FluxError(FLUX_E_IO, "cannot connect to " + endpoint + ": " + last);
    int one = 1;
    ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    return std::unique_ptr<TcpStream>(new TcpStream(fd));
}

// =================================================================================================
// NodeServer: flux-node/src/main.rs:21-111
// =================================================================================================
NodeServer::NodeServer(const std::string &host, const std::string &port, WorkerHandle worker, size_t num_threads)
    : worker_(std::move(worker)), num_threads_(num_threads) {
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    hints.ai_flags = AI_PASSIVE;
    const int rc = ::getaddrinfo(host.c_str(), port.c_str(), &hints, &res);
    if (rc != 0) throw FluxError(FLUX_E_IO, "cannot resolve bind address " + host + ":" + port + ": " + gai_strerror(rc));
    listen_fd_ = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);
    int one = 1;
    if (listen_fd_ >= 0) ::setsockopt(listen_fd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    if (listen_fd_ < 0 || ::bind(listen_fd_, res->ai_addr, res->ai_addrlen) != 0 || ::listen(listen_fd_, 4) != 0) {
        const std::string why = std::strerror(errno);
        ::freeaddrinfo(res);
        if (listen_fd_ >= 0) ::close(listen_fd_);
        listen_fd_ = -1;
        throw FluxError(FLUX_E_IO, "cannot listen on " + host + ":" + port + ": " + why);
    }
    ::freeaddrinfo(res);
    sockaddr_in sa{};
    socklen_t len = sizeof sa;
    if (::getsockname(listen_fd_, (sockaddr *)&sa, &len) == 0) port_ = ntohs(sa.sin_port);
}

NodeServer::~NodeServer() { stop(); }

void NodeServer::stop() {
    stopping_ = true;
    if (listen_fd_ >= 0) {
        ::shutdown(listen_fd_, SHUT_RDWR);
        ::close(listen_fd_);
        listen_fd_ = -1;
    }
}

void NodeServer::serve_forever() {  // run_server: one client at a time (main.rs:100-108)
    while (!stopping_) {
        const int lfd = listen_fd_;
        if (lfd < 0) break;
        const int fd = ::accept(lfd, nullptr, nullptr);
        if (fd < 0) {
            if (stopping_) break;
            if (errno == EINTR) continue;
            break;
        }
        int one = 1;
        ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        if (!handle_client(std::unique_ptr<TcpStream>(new TcpStream(fd))))
            std::printf("run_server: handle_client exited with an error\n");
        clients_++;
    }
}

bool NodeServer::handle_client(std::unique_ptr<TcpStream> stream) {
    std::printf("Got connection from %s\n", stream->peer().c_str());
    {
        Encoder e;  // main.rs:26-31
        encode_worker_info(e, WorkerInfo{num_threads_});
        if (!stream->write_all(e.out)) return false;
    }
    auto wu = std::make_shared<Channel<WorkUnit>>();
    auto re = std::make_shared<Channel<std::optional<RenderEvent>>>();
    auto wg = std::make_shared<WaitGroup>();
    TcpStream *raw = stream.get();
    // result writer (main.rs:41-55): every RenderEvent of the local worker goes back to the manager
    std::thread writer([raw, re] {
        while (auto m = re->recv()) {
            if (!*m) break;
            Encoder e;
            encode_event(e, **m);
            if (!raw->write_all(e.out)) {
                std::printf("Manager connection error\n");
                return;
            }
        }
    });
    bool ok = true;
    size_t jobs = 0;
    Decoder dec(*stream);
    for (;;) {  // for result in stream_de (main.rs:57-87)
        if (dec.peek() == Type::End) break;  // the client closed the stream
        NetworkWorkerRequest req;
        if (!decode_request(dec, req)) {
            std::printf("handle_client: bad request: %s\n", dec.error().empty() ? "unexpected message" : dec.error().c_str());
            ok = false;
            break;
        }
        if (req.kind == NetworkWorkerRequest::SetJob) {
            std::printf("Got job\n");
            wg->add();
            jobs++;
            worker_.send(std::make_shared<Job>(std::move(req.job)), wu, re, wg);
        } else if (req.kind == NetworkWorkerRequest::WorkUnitMsg) {
            wu->send(req.unit);
        } else {
            std::printf("Got done message, shutting down\n");
            break;
        }
    }
    // The reference returns right away and lets the channel drops end the worker's job; here the unit channel is
    // closed explicitly, the worker finishes the units it already holds, and its last events are flushed.
    wu->close();
    if (jobs) wg->wait();
    re->send(std::nullopt);
    writer.join();
    return ok;
}

// =================================================================================================
// NetworkWorker: workers.rs:112-258
// =================================================================================================
NetworkWorker::NetworkWorker(const std::string &raw_endpoint)
    : sender_(std::make_shared<Channel<std::optional<WorkerRequest>>>()) {
    stream_ = TcpStream::connect(raw_endpoint);
    std::printf("Getting info\n");
    Decoder d(*stream_);  // the first message is the node's WorkerInfo (workers.rs:134-141)
    if (!decode_worker_info(d, info_)) throw FluxError(FLUX_E_IO, "Could not get info from network node " + raw_endpoint);
    std::printf("Got info\n");
    thread_ = std::thread([this] { run(); });
}

NetworkWorker::~NetworkWorker() { stop(); }

void NetworkWorker::stop() {  // workers.rs:250-253
    if (stopped_) return;
    stopped_ = true;
    sender_->send(std::nullopt);
    if (thread_.joinable()) thread_.join();
}

void NetworkWorker::run() {
    Decoder events(*stream_);
    auto send_request = [&](const NetworkWorkerRequest &r) {
        Encoder e;
        encode_request(e, r);
        return stream_->write_all(e.out);
    };
    // forwards one RenderEvent from the node; false when the stream ended or broke
    auto forward_one = [&](const std::shared_ptr<Channel<std::optional<RenderEvent>>> &out) {
        RenderEvent ev;
        if (events.peek() == Type::End || !decode_event(events, ev)) return false;
        out->send(std::move(ev));
        return true;
    };
    for (;;) {  // while let Ok(Some((job, recv_unit, send_result, wg))) = r.recv()   (workers.rs:152)
        auto msg = sender_->recv();
        if (!msg || !*msg) break;
        WorkerRequest req = std::move(**msg);
        bool alive = true;
        NetworkWorkerRequest set;
        set.kind = NetworkWorkerRequest::SetJob;
        set.job = *req.job;
        alive = send_request(set);  // workers.rs:159
        size_t sent = 0;
        for (int k = 0; alive && k < 2; k++) {  // two units in flight (workers.rs:161-175)
            auto unit = req.recv_unit->recv();
            if (!unit) break;
            NetworkWorkerRequest r;
            r.kind = NetworkWorkerRequest::WorkUnitMsg;
            r.unit = *unit;
            alive = send_request(r);
            sent++;
        }
        while (alive) {  // one out, one in (workers.rs:179-201)
            auto unit = req.recv_unit->recv();
            if (!unit) break;
            NetworkWorkerRequest r;
            r.kind = NetworkWorkerRequest::WorkUnitMsg;
            r.unit = *unit;
            alive = send_request(r) && forward_one(req.send_result);
        }
        for (size_t k = 0; alive && k < sent; k++) alive = forward_one(req.send_result);  // workers.rs:205-222
        if (alive) {
            NetworkWorkerRequest done;
            done.kind = NetworkWorkerRequest::Done;  // workers.rs:226
            alive = send_request(done);
        }
        req.wg->done();  // drop(wg)
        if (!alive) {
            std::fprintf(stderr, "NetworkWorker: connection to the node lost\n");
            break;  // the reference's thread returns on a deserializer error (workers.rs:194-197)
        }
    }
}

}  // namespace flux_host
