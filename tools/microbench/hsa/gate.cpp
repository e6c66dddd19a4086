// Does a queue that is PARKED on a barrier-AND packet (waiting for a host-set signal) react sooner than an idle queue reacts to its doorbell?
// The fenced frame loop leaves each lane's queue idle for ~9 us between frames; a frame's chain starts with doorbell -> packet processor.
//   A: idle queue, barrier packet with a completion signal: doorbell -> host sees the signal, after `idle` us of nothing
//   B: the queue parked on a gate packet (barrier-AND, dep = gate signal at 1); after `idle` us the host appends a barrier packet with a completion
//      signal, rings the doorbell, opens the gate (signal store 0): append -> host sees the signal
//   C: as B, the second packet appended BEFORE the idle time (everything queued; only the gate store is timed): the gate's own latency
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); printf("%s: %s\n", #x, m_ ? m_ : "?"); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void spin(double us) { const double t = now(); while ((now() - t) * 1e6 < us) {} }
static hsa_agent_t g_gpu{};
static hsa_status_t pick_gpu(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && g_gpu.handle == 0) g_gpu = a;
    return HSA_STATUS_SUCCESS;
}
static hsa_queue_t* q; static uint64_t widx = 0;
static void barrier(hsa_signal_t dep, hsa_signal_t done) {
    auto* p = reinterpret_cast<hsa_barrier_and_packet_t*>(q->base_address) + (widx & (q->size - 1));
    memset((char*)p + 4, 0, 60);
    p->dep_signal[0] = dep; p->completion_signal = done;
    const uint16_t header = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)header, __ATOMIC_RELEASE);
    widx++;
    hsa_queue_store_write_index_screlease(q, widx);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(widx - 1));
}
static bool wait0(hsa_signal_t s, double limit_s = 2.0) { const double t = now(); while (hsa_signal_load_scacquire(s) != 0) if (now() - t > limit_s) return false; return true; }
static void report(const char* what, double idle, std::vector<double>& v) {
    std::sort(v.begin(), v.end());
    double m = 0; for (double x : v) m += x; m /= v.size();
    printf("%-64s idle %5.0f us: mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us\n", what, idle, m, v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
}
int main(int argc, char** argv) {
    HK(hsa_init());
    HK(hsa_iterate_agents(pick_gpu, nullptr));
    HK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    if (argc > 1) { const int pr = atoi(argv[1]); HK(hsa_amd_queue_set_priority(q, (hsa_amd_queue_priority_t)pr)); printf("queue priority %d (0 low, 1 normal, 2 high)\n", pr); }
    hsa_signal_t done, gate, none{}; HK(hsa_signal_create(1, 0, nullptr, &done)); HK(hsa_signal_create(1, 0, nullptr, &gate));
    const int N = 1500;
    for (double idle : {0.0, 10.0}) {
        std::vector<double> a, b, c;
        for (int i = 0; i < N; i++) {            // A
            spin(idle);
            hsa_signal_store_relaxed(done, 1);
            const double t0 = now();
            barrier(none, done);
            if (!wait0(done)) { printf("A: timeout\n"); return 1; }
            a.push_back((now() - t0) * 1e6);
        }
        for (int i = 0; i < N; i++) {            // B
            hsa_signal_store_relaxed(gate, 1); hsa_signal_store_relaxed(done, 1);
            barrier(gate, none);                 // park the queue
            spin(idle + 3.0);                    // (the packet processor has reached the gate)
            const double t0 = now();
            barrier(none, done);
            hsa_signal_store_screlease(gate, 0);
            if (!wait0(done)) { printf("B: timeout\n"); return 1; }
            b.push_back((now() - t0) * 1e6);
        }
        for (int i = 0; i < N; i++) {            // C
            hsa_signal_store_relaxed(gate, 1); hsa_signal_store_relaxed(done, 1);
            barrier(gate, none);
            barrier(none, done);
            spin(idle + 3.0);
            const double t0 = now();
            hsa_signal_store_screlease(gate, 0);
            if (!wait0(done)) { printf("C: timeout\n"); return 1; }
            c.push_back((now() - t0) * 1e6);
        }
        report("A idle queue: doorbell -> completion seen", idle, a);
        report("B parked queue: append + doorbell + open gate -> completion seen", idle, b);
        report("C parked queue, packet already queued: open gate -> completion seen", idle, c);
    }
    hsa_queue_destroy(q);
    return 0;
}
