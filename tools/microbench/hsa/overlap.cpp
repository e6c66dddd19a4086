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
    std::ifstream f(argc > 1 ? argv[1] : "k2.hsaco", std::ios::binary);
    std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (blob.empty()) { printf("k.hsaco not found\n"); return 1; }
    hsa_code_object_reader_t reader; HK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
    hsa_executable_t exe; HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HK(hsa_executable_freeze(exe, nullptr));
    hsa_executable_symbol_t sym; HK(hsa_executable_get_symbol_by_name(exe, "geo_kernel.kd", &g_gpu, &sym));
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
                                (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
        widx++;
        hsa_queue_store_write_index_relaxed(q, widx);
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(widx - 1));
    };

    // second kernel of the pair
    hsa_executable_symbol_t sym2; HK(hsa_executable_get_symbol_by_name(exe, "ras_kernel.kd", &g_gpu, &sym2));
    uint64_t kobj2; uint32_t lds2;
    HK(hsa_executable_symbol_get_info(sym2, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj2));
    HK(hsa_executable_symbol_get_info(sym2, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &lds2));
    struct GArgs { uint32_t* done; uint32_t* scratch; uint32_t spin; uint32_t pad; };
    struct RArgs { uint32_t* done; uint32_t* frame; uint32_t* stats; uint32_t target; uint32_t wait; };
    uint32_t *done, *gscratch, *frame, *stats;
    CK(hipMalloc(&done, 4096)); CK(hipMemset(done, 0, 4096)); CK(hipMalloc(&gscratch, 1 << 20)); CK(hipMalloc(&frame, 2040 * 256 * 4)); CK(hipMalloc(&stats, 64)); CK(hipMemset(stats, 0, 64));
    CK(hipDeviceSynchronize());
    auto put = [&](uint64_t ko, uint32_t ldsb, uint32_t gx, uint32_t gy, uint32_t bx, const void* args, size_t nbytes, bool barrier, hsa_signal_t done_sig) {
        while (widx - hsa_queue_load_read_index_relaxed(q) >= q->size - 8) {}
        uint8_t* ka = ring + (widx % SLOTS) * SLOT;
        Implicit im; memset(&im, 0, sizeof im);
        im.block_count[0] = gx; im.block_count[1] = gy; im.block_count[2] = 1; im.group_size[0] = (uint16_t)bx; im.group_size[1] = 1; im.group_size[2] = 1; im.grid_dims = 2;
        uint8_t tmp[SLOT]; memcpy(tmp, args, nbytes); memcpy(tmp + ((nbytes + 7) & ~7u), &im, 96); memcpy(ka, tmp, ((nbytes + 7) & ~7u) + 96);
        hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (widx & (q->size - 1));
        p->setup = 2 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p->workgroup_size_x = (uint16_t)bx; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->grid_size_x = gx * bx; p->grid_size_y = gy; p->grid_size_z = 1;
        p->private_segment_size = 0; p->group_segment_size = ldsb; p->kernel_object = ko; p->kernarg_address = ka; p->reserved2 = 0; p->completion_signal = done_sig;
        const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
        widx++;
        hsa_queue_store_write_index_relaxed(q, widx);
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(widx - 1));
    };
    const uint32_t GW = 157;
    uint32_t target = 0;
    for (uint32_t spin : {20u, 40u, 80u}) {
        for (int overlap = 0; overlap < 2; overlap++) {
            const int m = 200;
            CK(hipMemset(stats, 0, 64)); CK(hipDeviceSynchronize());
            hsa_signal_t none{0};
            double t0 = now();
            for (int i = 0; i < m; i++) {
                target += GW;
                GArgs ga{done, gscratch, spin, 0};
                RArgs ra{done, frame, stats, target, (uint32_t)overlap};
                put(kobj, lds, GW, 1, 64, &ga, sizeof ga, true, none);
                if (i == m - 1) { hsa_signal_store_relaxed(sig[0], 1); put(kobj2, lds2, 60, 34, 256, &ra, sizeof ra, overlap == 0, sig[0]); }
                else put(kobj2, lds2, 60, 34, 256, &ra, sizeof ra, overlap == 0, none);
            }
            while (hsa_signal_load_scacquire(sig[0]) != 0) {}
            const double dt = now() - t0;
            uint32_t st[2]; CK(hipMemcpy(st, stats, 8, hipMemcpyDeviceToHost));
            printf("geometry spin %u, raster packet %s: %.2f us per frame (timeouts %u, workgroups that waited %u of %d)\n", spin,
                   overlap ? "WITHOUT barrier bit, waits on the done counters" : "with barrier bit", 1e6 * dt / m, st[0], st[1], m * 2040);
            fflush(stdout);
        }
    }
    printf("ok\n");
    return 0;
}
