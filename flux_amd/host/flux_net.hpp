// flux_net.hpp -- the reference's render-node protocol: serde_cbor messages over one
// blocking TCP stream per node, default port 2000 (constants.rs:6).
//
//   server -> client, once : WorkerInfo{num_threads}                                   flux-node/src/main.rs:26-31
//   client -> server       : NetworkWorkerRequest::{SetJob(Box<Job>), WorkUnit(WorkUnit), Done}   workers.rs:105-110
//   server -> client       : a stream of RenderEvent (in practice RowsReady)            flux-node/src/main.rs:41-55
//
// NodeServer is flux-node's handle_client / run_server (flux-node/src/main.rs:21-111) in front of any Worker
// (the GPU worker in the flux_node binary); NetworkWorker is the client side (workers.rs:112-258): it keeps two
// work units in flight per node, then one-out-one-in, drains, and sends Done.
// Encoding: cbor.hpp (serde_cbor 0.9 layout; unverified against the real crate here).
#pragma once
#include <atomic>
#include <string>

#include "cbor.hpp"
#include "flux_host.hpp"

namespace flux_host {

constexpr const char *kDefaultPort = "2000";  // constants.rs:6

// NetworkWorkerRequest (workers.rs:105-110)
struct NetworkWorkerRequest {
    enum Kind { SetJob, WorkUnitMsg, Done } kind = Done;
    Job job;
    WorkUnit unit;
};

// ---- message codecs (serde derive layouts of job.rs, scene.rs, shapes.rs, color.rs, manager.rs) ----------
void encode_worker_info(cbor::Encoder &e, const WorkerInfo &w);
void encode_request(cbor::Encoder &e, const NetworkWorkerRequest &r);
void encode_event(cbor::Encoder &e, const RenderEvent &ev);
bool decode_worker_info(cbor::Decoder &d, WorkerInfo &w);
bool decode_request(cbor::Decoder &d, NetworkWorkerRequest &r);
bool decode_event(cbor::Decoder &d, RenderEvent &ev);

// ---- TCP ------------------------------------------------------------------------------------------
class TcpStream : public cbor::Reader {
public:
    TcpStream() = default;
    explicit TcpStream(int fd) : fd_(fd) {}
    ~TcpStream() override;
    TcpStream(const TcpStream &) = delete;
    TcpStream &operator=(const TcpStream &) = delete;
    // "host" or "host:port" (workers.rs:120-123); throws FluxError(FLUX_E_IO) on failure
    static std::unique_ptr<TcpStream> connect(const std::string &endpoint);
    bool read(void *dst, size_t n) override;
    bool write_all(const std::string &bytes);
    void shutdown_both();
    std::string peer() const;
    int fd() const { return fd_; }
private:
    int fd_ = -1;
};

// flux-node: bind, accept one client at a time, serve it with `worker` (flux-node/src/main.rs:96-111).
class NodeServer {
public:
    NodeServer(const std::string &host, const std::string &port, WorkerHandle worker, size_t num_threads);
    ~NodeServer();
    uint16_t port() const { return port_; }       // the bound port (useful with port "0")
    void serve_forever();                          // run_server; returns after stop()
    void stop();                                   // closes the listener (unblocks accept)
    size_t clients_served() const { return clients_; }
private:
    bool handle_client(std::unique_ptr<TcpStream> stream);  // handle_client, main.rs:21-94
    int listen_fd_ = -1;
    uint16_t port_ = 0;
    WorkerHandle worker_;
    size_t num_threads_;
    std::atomic<bool> stopping_{false};
    std::atomic<size_t> clients_{0};
};

// NetworkWorker (workers.rs:112-258)
class NetworkWorker : public Worker {
public:
    explicit NetworkWorker(const std::string &raw_endpoint);  // connects and reads WorkerInfo; throws on failure
    ~NetworkWorker() override;
    WorkerHandle handle() const override { return WorkerHandle(sender_); }
    void stop() override;
    WorkerInfo info() const override { return info_; }
private:
    void run();
    std::unique_ptr<TcpStream> stream_;
    WorkerInfo info_;
    std::shared_ptr<Channel<std::optional<WorkerRequest>>> sender_;
    std::thread thread_;
    bool stopped_ = false;
};

}  // namespace flux_host
