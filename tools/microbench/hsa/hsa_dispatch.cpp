// Prototype: dispatching a kernel by writing the AQL packet ourselves (ROCr / HSA user-mode queue) next to a live HIP runtime --
// what does a launch cost the host then, and how soon does the host see it finish?  (HIP: 2.3 - 3.0 us per launch whatever the
// entry point, round trip 6.4 - 9.5 us: launch_paths.hip, fence_latency.hip.)
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); printf("%s: %s\n", #x, m_ ? m_ : "?"); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Args { uint32_t* out; uint32_t value; uint32_t pad; uint32_t* word; };
// code object v5 implicit arguments, as far as blockIdx / gridDim / blockDim need them
struct Implicit { uint32_t block_count[3]; uint16_t group_size[3]; uint16_t remainder[3]; uint8_t reserved[16]; uint64_t global_offset[3]; uint16_t grid_dims; uint8_t rest[190]; };
static_assert(sizeof(Implicit) == 256, "implicit block");

static hsa_agent_t g_gpu{};
static hsa_status_t pick_gpu(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && g_gpu.handle == 0) g_gpu = a;
    return HSA_STATUS_SUCCESS;
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    void* warm; CK(hipMalloc(&warm, 4096));            // HIP is up (and has initialised ROCr)
    HK(hsa_init());
    HK(hsa_iterate_agents(pick_gpu, nullptr));
    char name[64]; hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, name); printf("agent: %s\n", name);
    // code object
    std::ifstream f(argc > 1 ? argv[1] : "k.hsaco", std::ios::binary);
    std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (blob.empty()) { printf("k.hsaco not found\n"); return 1; }
    hsa_code_object_reader_t reader; HK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
    hsa_executable_t exe; HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HK(hsa_executable_freeze(exe, nullptr));
    hsa_executable_symbol_t sym; HK(hsa_executable_get_symbol_by_name(exe, "probe_kernel.kd", &g_gpu, &sym));
    uint64_t kobj; uint32_t karg, lds, scratch;
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &karg));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &lds));
    HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &scratch));
    printf("kernel object %#lx, kernarg %u B (explicit %zu), LDS %u, scratch %u\n", kobj, karg, sizeof(Args), lds, scratch);
    // queue
    hsa_queue_t* q; HK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    // kernarg ring in fine-grained device memory (host-written over the BAR), completion signals, output
    const uint32_t SLOT = 512, SLOTS = 256;
    uint8_t* ring; CK(hipExtMallocWithFlags((void**)&ring, SLOT * SLOTS, hipDeviceMallocFinegrained));
    uint32_t* out; CK(hipMalloc(&out, 4096 * 4)); CK(hipMemset(out, 0, 4096 * 4));
    volatile uint32_t* word; CK(hipHostMalloc((void**)&word, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    uint32_t* word_dev; CK(hipHostGetDevicePointer((void**)&word_dev, (void*)word, 0));
    hsa_signal_t sig[4]; for (auto& s : sig) HK(hsa_signal_create(1, 0, nullptr, &s));
    uint64_t widx = 0;
    uint32_t acq_scope = HSA_FENCE_SCOPE_SYSTEM, rel_scope = HSA_FENCE_SCOPE_SYSTEM;
    auto dispatch = [&](uint32_t gx, uint32_t gy, uint32_t value, uint32_t* w, hsa_signal_t done, bool barrier) {
        uint8_t* ka = ring + (widx % SLOTS) * SLOT;
        Args a{out, value, 0, w};
        Implicit im; memset(&im, 0, sizeof im);
        im.block_count[0] = gx; im.block_count[1] = gy; im.block_count[2] = 1; im.group_size[0] = 256; im.group_size[1] = 1; im.group_size[2] = 1; im.grid_dims = 2;
        uint8_t tmp[SLOT]; memcpy(tmp, &a, sizeof a); memcpy(tmp + ((sizeof a + 7) & ~7u), &im, 96);
        memcpy(ka, tmp, ((sizeof a + 7) & ~7u) + 96);
        hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (widx & (q->size - 1));
        p->setup = 2 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->grid_size_x = gx * 256; p->grid_size_y = gy; p->grid_size_z = 1;
        p->private_segment_size = scratch; p->group_segment_size = lds;
        p->kernel_object = kobj; p->kernarg_address = ka; p->reserved2 = 0; p->completion_signal = done;
        const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                (acq_scope << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel_scope << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
        widx++;
        hsa_queue_store_write_index_relaxed(q, widx);
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(widx - 1));
    };
    // correctness: a 60 x 34 grid
    hsa_signal_store_relaxed(sig[0], 1);
    dispatch(60, 34, 1000, nullptr, sig[0], true);
    while (hsa_signal_load_scacquire(sig[0]) != 0) {}
    static uint32_t h[4096]; CK(hipMemcpy(h, out, 2040 * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (uint32_t i = 0; i < 2040; i++) bad += h[i] != 1000 + i + (60u << 16) + (34u << 24);
    printf("2040 workgroups, blockIdx / gridDim through the implicit arguments: %d wrong\n", bad);
    // host cost per dispatch (no completion signal), then drain
    const int n = 20000;
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        for (int i = 0; i < n; i++) {
            while (widx - hsa_queue_load_read_index_relaxed(q) >= q->size - 8) {}
            hsa_signal_t none{0};
            if (i == n - 1) { hsa_signal_store_relaxed(sig[1], 1); dispatch(1, 1, 5, nullptr, sig[1], true); } else dispatch(1, 1, 5, nullptr, none, true);
        }
        double t1 = now();
        while (hsa_signal_load_scacquire(sig[1]) != 0) {}
        printf("AQL dispatch by hand: host %.3f us per dispatch, %.3f us incl. drain\n", 1e6 * (t1 - t0) / n, 1e6 * (now() - t0) / n);
    }
    // where the 2.5 us go: kernarg stores (BAR), packet stores, the doorbell -- and two packets behind ONE doorbell (a frame's two kernels)
    {
        double t_ka = 0, t_pk = 0, t_db = 0;
        const int m = 20000;
        for (int i = 0; i < m; i++) {
            while (widx - hsa_queue_load_read_index_relaxed(q) >= q->size - 8) {}
            uint8_t* ka = ring + (widx % SLOTS) * SLOT;
            Args a{out, 5, 0, nullptr};
            Implicit im; memset(&im, 0, sizeof im);
            im.block_count[0] = 1; im.block_count[1] = 1; im.block_count[2] = 1; im.group_size[0] = 256; im.group_size[1] = 1; im.group_size[2] = 1; im.grid_dims = 2;
            double a0 = now();
            uint8_t tmp[SLOT]; memcpy(tmp, &a, sizeof a); memcpy(tmp + 24, &im, 96); memcpy(ka, tmp, 120);
            double a1 = now();
            hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (widx & (q->size - 1));
            p->setup = 2 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->grid_size_x = 256; p->grid_size_y = 1; p->grid_size_z = 1;
            p->private_segment_size = scratch; p->group_segment_size = lds; p->kernel_object = kobj; p->kernarg_address = ka; p->reserved2 = 0; p->completion_signal.handle = 0;
            const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                    (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
            __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
            widx++;
            hsa_queue_store_write_index_relaxed(q, widx);
            double a2 = now();
            hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(widx - 1));
            double a3 = now();
            t_ka += a1 - a0; t_pk += a2 - a1; t_db += a3 - a2;
        }
        printf("per dispatch: kernarg stores %.3f us, packet stores %.3f us, doorbell %.3f us (each incl. ~0.03 us of clock reads)\n", 1e6 * t_ka / m, 1e6 * t_pk / m, 1e6 * t_db / m);
        hsa_signal_store_relaxed(sig[3], 1);
        dispatch(1, 1, 5, nullptr, sig[3], true);
        while (hsa_signal_load_scacquire(sig[3]) != 0) {}
        // pairs: two packets, one doorbell
        double t0 = now();
        for (int i = 0; i < m; i++) {
            while (widx - hsa_queue_load_read_index_relaxed(q) >= q->size - 8) {}
            for (int k = 0; k < 2; k++) {
                uint8_t* ka = ring + (widx % SLOTS) * SLOT;
                Args a{out, 5, 0, nullptr};
                Implicit im; memset(&im, 0, sizeof im);
                im.block_count[0] = 1; im.block_count[1] = 1; im.block_count[2] = 1; im.group_size[0] = 256; im.group_size[1] = 1; im.group_size[2] = 1; im.grid_dims = 2;
                uint8_t tmp[SLOT]; memcpy(tmp, &a, sizeof a); memcpy(tmp + 24, &im, 96); memcpy(ka, tmp, 120);
                hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (widx & (q->size - 1));
                p->setup = 2 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
                p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->grid_size_x = 256; p->grid_size_y = 1; p->grid_size_z = 1;
                p->private_segment_size = scratch; p->group_segment_size = lds; p->kernel_object = kobj; p->kernarg_address = ka; p->reserved2 = 0; p->completion_signal.handle = 0;
                const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                        (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
                __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
                widx++;
            }
            hsa_queue_store_write_index_relaxed(q, widx);
            hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(widx - 1));
        }
        double t1 = now();
        hsa_signal_store_relaxed(sig[3], 1);
        dispatch(1, 1, 5, nullptr, sig[3], true);
        while (hsa_signal_load_scacquire(sig[3]) != 0) {}
        printf("two packets behind one doorbell: host %.3f us per PAIR, %.3f us incl. drain\n", 1e6 * (t1 - t0) / m, 1e6 * (now() - t0) / m);
    }
    // round trips: completion signal polled; a pinned word written by the kernel polled
    {
        const int m = 5000;
        double t0 = now();
        for (int i = 1; i <= m; i++) {
            hsa_signal_store_relaxed(sig[2], 1);
            dispatch(1, 1, (uint32_t)i, nullptr, sig[2], true);
            while (hsa_signal_load_scacquire(sig[2]) != 0) {}
        }
        printf("round trip, completion signal polled : %.2f us\n", 1e6 * (now() - t0) / m);
        hsa_signal_t none{0};
        t0 = now();
        for (int i = 1; i <= m; i++) {
            dispatch(1, 1, (uint32_t)i, word_dev, none, true);
            while (*word != (uint32_t)i) {}
        }
        printf("round trip, kernel-written pinned word: %.2f us\n", 1e6 * (now() - t0) / m);
        // what the completion-signal path is made of: fence scopes of the packet, and a signal without an interrupt mailbox
        hsa_signal_t gpu_only; HK(hsa_amd_signal_create(1, 0, nullptr, HSA_AMD_SIGNAL_AMD_GPU_ONLY, &gpu_only));
        const char* sn[3] = {"none", "agent", "system"};
        for (int variant = 0; variant < 2; variant++)
            for (uint32_t acq = 0; acq < 3; acq++)
                for (uint32_t rel = 0; rel < 3; rel++) {
                    hsa_signal_t sg = variant ? gpu_only : sig[2];
                    acq_scope = acq; rel_scope = rel;
                    t0 = now();
                    for (int i = 1; i <= m; i++) {
                        hsa_signal_store_relaxed(sg, 1);
                        dispatch(1, 1, (uint32_t)i, nullptr, sg, true);
                        while (hsa_signal_load_scacquire(sg) != 0) {}
                    }
                    printf("round trip, %s signal, acquire %s release %s: %.2f us\n", variant ? "GPU-only (no interrupt)" : "default", sn[acq], sn[rel], 1e6 * (now() - t0) / m);
                }
        acq_scope = rel_scope = HSA_FENCE_SCOPE_SYSTEM;
        // the same with a 2040-workgroup kernel (a frame's raster kernel): does the release scope matter when there is something to write back?
        for (uint32_t rel = 1; rel < 3; rel++) {
            rel_scope = rel;
            t0 = now();
            for (int i = 1; i <= m; i++) {
                hsa_signal_store_relaxed(gpu_only, 1);
                dispatch(60, 34, (uint32_t)i, nullptr, gpu_only, true);
                while (hsa_signal_load_scacquire(gpu_only) != 0) {}
            }
            printf("round trip, 2040 workgroups, GPU-only signal, release %s: %.2f us\n", sn[rel], 1e6 * (now() - t0) / m);
            t0 = now();
            for (int i = 1; i <= m; i++) {
                hsa_signal_store_relaxed(sig[2], 1);
                dispatch(60, 34, (uint32_t)i, nullptr, sig[2], true);
                while (hsa_signal_load_scacquire(sig[2]) != 0) {}
            }
            printf("round trip, 2040 workgroups, default signal, release %s: %.2f us\n", sn[rel], 1e6 * (now() - t0) / m);
        }
        rel_scope = HSA_FENCE_SCOPE_SYSTEM;
        // two-deep pipeline with a 2040-workgroup kernel, completion signals
        t0 = now();
        for (int i = 1; i <= m; i++) {
            if (i > 2) while (hsa_signal_load_scacquire(sig[i & 1]) != 0) {}
            hsa_signal_store_relaxed(sig[i & 1], 1);
            dispatch(60, 34, (uint32_t)i, nullptr, sig[i & 1], true);
        }
        while (hsa_signal_load_scacquire(sig[0]) != 0 || hsa_signal_load_scacquire(sig[1]) != 0) {}
        printf("two in flight, 2040-workgroup kernel, completion signals: %.2f us per launch\n", 1e6 * (now() - t0) / m);
    }
    // HIP still works beside it
    CK(hipMemset(out, 0, 64)); CK(hipDeviceSynchronize());
    printf("ok\n");
    return 0;
}
