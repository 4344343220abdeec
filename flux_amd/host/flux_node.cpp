// flux_node.cpp -- the reference's flux-node daemon (flux-node/src/main.rs) with a GPU worker behind it:
// binds host:port, serves one manager connection at a time, renders the work units it is sent on one
// MI355X and streams RenderEvent::RowsReady back (protocol: flux_net.hpp).
// Flags (main.rs:119-152): -h/--host ADDRESS [0.0.0.0], -p/--port PORT [2000], -t/--threads N (reported in
// WorkerInfo; default: 1 -- one GPU worker).  Additions: --device I [0], --seed S [1], --once (exit after the
// first client; used by the tests).  `--help` prints usage (-h is the host flag, as in the reference).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>

#include "flux_host.hpp"
#include "flux_net.hpp"

using namespace flux_host;

int main(int argc, char **argv) {
    std::string host = "0.0.0.0", port = kDefaultPort;
    size_t threads = 1;
    int device = 0;
    uint64_t seed = 1;
    bool once = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](const char *flag) -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "error: The argument '%s' requires a value\n", flag);
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "-h" || a == "--host") host = need("--host <ADDRESS>");
        else if (a == "-p" || a == "--port") port = need("--port <port>");
        else if (a == "-t" || a == "--threads") threads = (size_t)std::strtoull(need("--threads <threads>"), nullptr, 10);
        else if (a == "--device") device = std::atoi(need("--device <i>"));
        else if (a == "--seed") seed = std::strtoull(need("--seed <seed>"), nullptr, 10);
        else if (a == "--once") once = true;
        else if (a == "--help") {
            std::puts("flux-node\nNetwork rendering server for the flux ray tracer (MI355X render path)\n\nUSAGE:\n"
                      "    flux_node [OPTIONS]\n\nOPTIONS:\n    -h, --host <ADDRESS>    Listen for requests on this address\n"
                      "    -p, --port <port>       Listen on this TCP port\n    -t, --threads <threads> Reported in WorkerInfo\n"
                      "        --device <i>        GPU to render on\n        --seed <seed>\n        --once              serve one client, then exit");
            return 0;
        } else {
            std::fprintf(stderr, "error: Found argument '%s' which wasn't expected\n", a.c_str());
            return 2;
        }
    }
    if (flux_device_count() <= device) {
        std::fprintf(stderr, "error: no HIP device %d (this renderer has no CPU fallback)\n", device);
        return 1;
    }
    std::printf("Bind address: %s:%s\n", host.c_str(), port.c_str());  // main.rs:158
    GpuWorker worker(device, seed);
    try {
        NodeServer server(host, port, worker.handle(), threads);
        std::printf("Listening on port %u\n", (unsigned)server.port());
        std::fflush(stdout);
        if (once) {
            // serve exactly one client: stop the listener from a helper once a client has come and gone
            std::thread t([&] {
                while (server.clients_served() == 0) std::this_thread::sleep_for(std::chrono::milliseconds(20));
                server.stop();
            });
            server.serve_forever();
            t.join();
        } else {
            server.serve_forever();
        }
    } catch (const FluxError &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        worker.stop();
        return 1;
    }
    worker.stop();
    return 0;
}
