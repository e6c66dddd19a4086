// Which acquire scope does the FIRST packet of a frame need on our own AQL queue?  A kernel reads a buffer (its lines now sit in the L2 of
// every XCD), something that is not a kernel of ours rewrites the buffer, and the next packet's kernel reads it again with acquire scope
// none / agent / system: how many stale words does it see?  (hsa_dispatch.cpp: a system-scope acquire costs 2.6 us per packet.)
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <immintrin.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); printf("%s: %s\n", #x, m_ ? m_ : "?"); exit(1); } } while (0)
struct Args { const uint32_t* x; uint32_t n; uint32_t expect; uint32_t* bad; };
struct Implicit { uint32_t block_count[3]; uint16_t group_size[3]; uint16_t remainder[3]; uint8_t reserved[16]; uint64_t global_offset[3]; uint16_t grid_dims; uint8_t rest[190]; };
__global__ void hip_fill(uint32_t* x, uint32_t n, uint32_t v) { for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) x[i] = v; }
static hsa_agent_t g_gpu{};
static hsa_status_t pick_gpu(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && g_gpu.handle == 0) g_gpu = a;
    return HSA_STATUS_SUCCESS;
}
int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    HK(hsa_init());
    HK(hsa_iterate_agents(pick_gpu, nullptr));
    std::ifstream f(argc > 1 ? argv[1] : "k3.hsaco", std::ios::binary);
    std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (blob.empty()) { printf("k3.hsaco not found\n"); return 1; }
    hsa_code_object_reader_t reader; HK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
    hsa_executable_t exe; HK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HK(hsa_executable_freeze(exe, nullptr));
    hsa_executable_symbol_t sym; HK(hsa_executable_get_symbol_by_name(exe, "check_kernel.kd", &g_gpu, &sym));
    uint64_t kobj; HK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    hsa_queue_t* q; HK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    const uint32_t SLOT = 512, SLOTS = 256;
    uint8_t* ring; CK(hipExtMallocWithFlags((void**)&ring, SLOT * SLOTS, hipDeviceMallocFinegrained));
    hsa_signal_t sig; HK(hsa_signal_create(1, 0, nullptr, &sig));
    uint64_t widx = 0;
    auto dispatch = [&](const uint32_t* x, uint32_t n, uint32_t expect, uint32_t* bad, uint32_t acq, uint32_t rel = HSA_FENCE_SCOPE_SYSTEM) {
        uint8_t* ka = ring + (widx % SLOTS) * SLOT;
        Args a{x, n, expect, bad};
        Implicit im; memset(&im, 0, sizeof im);
        im.block_count[0] = 2040; im.block_count[1] = 1; im.block_count[2] = 1; im.group_size[0] = 256; im.group_size[1] = 1; im.group_size[2] = 1; im.grid_dims = 1;
        uint8_t tmp[SLOT]; memcpy(tmp, &a, sizeof a); memcpy(tmp + sizeof a, &im, 96); memcpy(ka, tmp, sizeof a + 96);
        _mm_sfence();
        hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (widx & (q->size - 1));
        p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1; p->grid_size_x = 2040 * 256; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = 0; p->group_segment_size = 0; p->kernel_object = kobj; p->kernarg_address = ka; p->reserved2 = 0; p->completion_signal = sig;
        const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        hsa_signal_store_relaxed(sig, 1);
        __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
        widx++;
        hsa_queue_store_write_index_relaxed(q, widx);
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(widx - 1));
        while (hsa_signal_load_scacquire(sig) != 0) {}
    };
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const char* sn[3] = {"none", "agent", "system"};
    const char* wn[7] = {"hipMemcpy from pageable host memory", "hipMemcpyAsync from pinned host memory + stream sync", "hipMemcpy device to device",
                         "hipMemsetAsync (D32) + stream sync", "a HIP kernel on a HIP stream + stream sync", "host stores over the BAR into fine-grained device memory + sfence",
                         "hipMemcpyAsync device to device on a stream + stream sync"};
    for (uint32_t n : {1024u, 65536u}) {
        uint32_t *x, *xf, *src, *pinned, *bad_dev;
        CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&src, n * 4));
        CK(hipExtMallocWithFlags((void**)&xf, n * 4, hipDeviceMallocFinegrained));
        CK(hipHostMalloc((void**)&pinned, n * 4, hipHostMallocDefault));
        CK(hipExtMallocWithFlags((void**)&bad_dev, 4096, hipDeviceMallocFinegrained));   // (fine-grained: the atomics are not held back in an L2)
        std::vector<uint32_t> pageable(n);
        printf("buffer of %u KB, 300 rewrites each:\n", n * 4 / 1024);
        for (int w = 0; w < 7; w++) {
            if (w == 5 && n > 16384) continue;
            for (uint32_t rel = 0; rel < 3; rel++)
            for (uint32_t acq = 0; acq < 3; acq++) {
                uint32_t* target = w == 5 ? xf : x;
                CK(hipMemset(target, 0, n * 4)); CK(hipMemset(bad_dev, 0, 4)); CK(hipDeviceSynchronize());
                dispatch(target, n, 0, bad_dev, HSA_FENCE_SCOPE_SYSTEM);          // lines into every L2
                uint64_t stale_words = 0; int stale_runs = 0;
                for (uint32_t it = 1; it <= 300; it++) {
                    switch (w) {
                        case 0: for (auto& v : pageable) v = it; CK(hipMemcpy(target, pageable.data(), n * 4, hipMemcpyHostToDevice)); break;
                        case 1: for (uint32_t i = 0; i < n; i++) pinned[i] = it; CK(hipMemcpyAsync(target, pinned, n * 4, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); break;
                        case 2: hip_fill<<<64, 256, 0, st>>>(src, n, it); CK(hipStreamSynchronize(st)); CK(hipMemcpy(target, src, n * 4, hipMemcpyDeviceToDevice)); CK(hipDeviceSynchronize()); break;
                        case 3: CK(hipMemsetD32Async((hipDeviceptr_t)target, (int)it, n, st)); CK(hipStreamSynchronize(st)); break;
                        case 4: hip_fill<<<64, 256, 0, st>>>(target, n, it); CK(hipStreamSynchronize(st)); break;
                        case 5: for (uint32_t i = 0; i < n; i++) ((volatile uint32_t*)target)[i] = it; _mm_sfence(); break;
                        case 6: hip_fill<<<64, 256, 0, st>>>(src, n, it); CK(hipMemcpyAsync(target, src, n * 4, hipMemcpyDeviceToDevice, st)); CK(hipStreamSynchronize(st)); break;
                    }
                    dispatch(target, n, it, bad_dev, acq, rel);     // (its release scope is what the NEXT rewrite finds the caches in)
                    uint32_t bad; CK(hipMemcpy(&bad, bad_dev, 4, hipMemcpyDeviceToHost));
                    if (bad) { stale_words += bad; stale_runs++; CK(hipMemset(bad_dev, 0, 4)); CK(hipDeviceSynchronize()); }
                }
                printf("  %-68s release %-6s acquire %-6s: %d of 300 reads saw stale words (%.0f word reads of %.0f)\n", wn[w], sn[rel], sn[acq], stale_runs, (double)stale_words, 300.0 * 2040 * n);
                fflush(stdout);
            }
        }
        CK(hipFree(x)); CK(hipFree(xf)); CK(hipFree(src)); CK(hipHostFree(pinned)); CK(hipFree(bad_dev));
    }
    printf("ok\n");
    return 0;
}
