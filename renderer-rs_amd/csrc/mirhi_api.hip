// mirhi_api.hip -- implementation of the C ABI declared in include/mirhi.h.
//
// Host-only logic: object lifetimes, validation with the reference's error behaviour
// (crates/rhi/src/{buffer,pipeline,command,sync}.rs), translation of a recorded command buffer
// into kernel launches on a HIP stream (crates/renderer/src/renderer.rs:452-557,
// frame_manager.rs:299-539).  No rasterization arithmetic lives here and there is no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/amd_hsa_signal.h>
#include <rccl/rccl.h>     // types and prototypes only: librccl is loaded with dlopen on first use (no link-time dependency)

#include "../../include/mirhi.h"
#include "mirhi_device.h"
#include "mirhi_launch.h"

using namespace mirhi;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static mirhi_result fail(mirhi_result code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
static mirhi_result hip_fail(hipError_t e, const char* what) {
    const mirhi_result code = (e == hipErrorOutOfMemory) ? MIRHI_ERR_ALLOCATOR : MIRHI_ERR_DEVICE;
    return fail(code, "%s: %s (%s)", code == MIRHI_ERR_ALLOCATOR ? "Allocator error" : "Vulkan error", hipGetErrorString(e), what);
}
#define HIP_TRY(expr)                                              \
    do {                                                           \
        hipError_t e__ = (expr);                                   \
        if (e__ != hipSuccess) return hip_fail(e__, #expr);        \
    } while (0)
#define NULL_CHECK(p, name)                                        \
    do {                                                           \
        if (!(p)) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: %s is null", name); \
    } while (0)

extern "C" const char* mirhi_last_error_message(void) { return g_last_error.c_str(); }
extern "C" uint32_t mirhi_abi_version(void) { return MIRHI_ABI_VERSION; }
extern "C" const char* mirhi_result_name(mirhi_result r) {
    switch (r) {
        case MIRHI_OK: return "Ok";
        case MIRHI_ERR_DEVICE: return "VulkanError";
        case MIRHI_ERR_LOADING: return "LoadingError";
        case MIRHI_ERR_ALLOCATOR: return "AllocatorError";
        case MIRHI_ERR_NO_SUITABLE_GPU: return "NoSuitableGpu";
        case MIRHI_ERR_SHADER: return "ShaderError";
        case MIRHI_ERR_SURFACE: return "SurfaceError";
        case MIRHI_ERR_SWAPCHAIN: return "SwapchainError";
        case MIRHI_ERR_INVALID_HANDLE: return "InvalidHandle";
        case MIRHI_ERR_PIPELINE: return "PipelineError";
        case MIRHI_ERR_LOCK_POISONED: return "LockPoisoned";
        case MIRHI_TIMEOUT: return "Timeout";
        case MIRHI_NOT_READY: return "NotReady";
        default: return "Unknown";
    }
}

static inline void cpu_relax() { __builtin_ia32_pause(); }

// ------------------------------------------------------------------------------------------------
// native dispatch: AQL packets written by this library (mirhi_native.h)
// ------------------------------------------------------------------------------------------------
#ifndef MIRHI_SOURCE_HASH
#define MIRHI_SOURCE_HASH "unknown"           // build.py passes the hash of csrc/ + include/mirhi.h to BOTH translation units
#endif
extern "C" const char* mirhi_build_id(void) { return MIRHI_SOURCE_HASH; }

namespace mirhi {
struct NativeKernel { uint64_t object; uint32_t kernarg_bytes, lds_bytes, scratch_bytes; };
struct NativeDevice {
    hsa_agent_t agent{};
    hsa_executable_t exe{};
    std::mutex mu;
    std::vector<std::pair<const void*, NativeKernel>> kernels;      // by host function address (a handful: linear search)
    bool ok = false;
    std::string why;
    std::atomic<uint64_t> dispatches{0};
    std::atomic<bool> lost{false};           // a wait on one of this device's queues ran into its deadline: nothing more is dispatched, every wait fails
    std::string lost_why;
    uint64_t timeout_ns = 10000000000ull;    // deadline of every native wait WITHOUT progress (MIRHI_NATIVE_TIMEOUT_MS)
};
struct NativeQueue {
    NativeDevice* nd = nullptr;
    hsa_queue_t* q = nullptr;
    uint8_t* ring = nullptr;                 // kernel arguments: fine-grained device memory, host-written
    size_t ring_bytes = 0;
    uint64_t widx = 0, drained = 0;          // packets written so far / packets known to have completed
    hsa_signal_t drain_sig{};
    uint64_t seen_foreign = 0;               // mirhi_device::foreign_writes at this queue's last system-scope acquire
    bool proxy = false;                      // hsa_queue_create handed out a software queue (a tool intercepts the packets): see native_queue_open
    std::mutex mu;                           // one producer at a time (a host thread that waits for the queue also writes a packet)
};
}  // namespace mirhi
namespace {
// code object v5 implicit kernel arguments as far as the kernels need them (blockIdx / gridDim / blockDim)
struct ImplicitArgs { uint32_t block_count[3]; uint16_t group_size[3]; uint16_t remainder[3]; uint8_t reserved[16]; uint64_t global_offset[3]; uint16_t grid_dims; uint8_t pad[6]; };
static_assert(sizeof(ImplicitArgs) == 72, "implicit argument block (the part this library fills)");
constexpr size_t NATIVE_RING_BYTES = 256 * 1024;
constexpr size_t NATIVE_KERNARG_SLOT = 1024;      // bytes of kernel arguments (explicit + implicit) a packet may have
constexpr uint32_t NATIVE_QUEUE_PACKETS = 1024;

// environment switches read ONCE (getenv walks the whole environment: ~0.1 us each, and a submit asked five of them)
struct NativeEnv {
    int dispatch = 1;            // MIRHI_NATIVE_DISPATCH: 0 off, 1 on where available, 2 required (a device that cannot start it fails to create)
    int system_scope = 0;        // MIRHI_NATIVE_SYSTEM_SCOPE: 1 system scope on every packet, 2 on every scope's first (A/B runs)
    int geom_tpw = 0;            // MIRHI_GEOM_TPW: triangles per geometry wave of small scopes (16 / 32 / 64; 0 = the host's choice)
    bool no_batch = false, fence_record = false;
    uint64_t timeout_ns = 10000000000ull;
    NativeEnv() {
        auto num = [](const char* n, int d) { const char* v = getenv(n); return v ? atoi(v) : d; };
        dispatch = num("MIRHI_NATIVE_DISPATCH", 1); system_scope = num("MIRHI_NATIVE_SYSTEM_SCOPE", 0);
        geom_tpw = num("MIRHI_GEOM_TPW", 0);
        no_batch = getenv("MIRHI_NO_BATCH") != nullptr; fence_record = getenv("MIRHI_FENCE_RECORD") != nullptr;
        const int ms = num("MIRHI_NATIVE_TIMEOUT_MS", 10000);
        timeout_ns = (uint64_t)(ms > 0 ? ms : 10000) * 1000000ull;
    }
};
const NativeEnv& native_env() { static const NativeEnv e; return e; }

struct PickAgent { uint32_t domain, bdf; hsa_agent_t found{}; uint32_t matches = 0; };
hsa_status_t native_pick_agent(hsa_agent_t a, void* data) {
    auto* want = static_cast<PickAgent*>(data);      // PCI domain and bus:device.function of the HIP device
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, domain = 0;
    if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain) != HSA_STATUS_SUCCESS) domain = 0;
    if ((bdf & 0xFFFFu) == (want->bdf & 0xFFFFu) && domain == want->domain) { if (want->matches++ == 0) want->found = a; }
    return HSA_STATUS_SUCCESS;
}

// the directory this shared library was loaded from (the code object sits beside it)
// <dir>/<library name without .so>_kernels.hsaco: the code object build.py took out of THIS library's object (libmirhi.so -> libmirhi_kernels.hsaco; a
// variant build beside it -- tools/ab_bench.sh -- finds its own)
std::string native_code_object_path() {
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void*>(&native_code_object_path), &info) || !info.dli_fname) return "./libmirhi_kernels.hsaco";
    std::string p = info.dli_fname;
    const size_t dot = p.rfind(".so");
    if (dot != std::string::npos) p.erase(dot);
    return p + "_kernels.hsaco";
}

mirhi::NativeDevice* native_device_open(int ordinal) {
    auto* nd = new mirhi::NativeDevice();
    auto fail_with = [&](const std::string& w) { nd->why = w; return nd; };
    nd->timeout_ns = native_env().timeout_ns;
    if (native_env().dispatch == 0) return fail_with("switched off (MIRHI_NATIVE_DISPATCH=0)");
    if (hsa_init() != HSA_STATUS_SUCCESS) return fail_with("hsa_init failed");
    int bus = 0, devid = 0, domain = 0;
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, ordinal) != hipSuccess || hipDeviceGetAttribute(&devid, hipDeviceAttributePciDeviceId, ordinal) != hipSuccess) { (void)hipGetLastError(); return fail_with("no PCI id for the HIP device"); }
    if (hipDeviceGetAttribute(&domain, hipDeviceAttributePciDomainID, ordinal) != hipSuccess) { (void)hipGetLastError(); domain = 0; }
    // the ROCr agent of the HIP ordinal: PCI domain AND bus:device.function (hosts with several PCI domains repeat bus numbers); an ambiguous match is refused
    PickAgent want; want.domain = (uint32_t)domain; want.bdf = (uint32_t)((bus << 8) | (devid << 3));
    (void)hsa_iterate_agents(native_pick_agent, &want);
    if (want.matches == 0) return fail_with("no ROCr agent with the HIP device's PCI address");
    if (want.matches > 1) return fail_with("several ROCr agents share the HIP device's PCI address");
    nd->agent = want.found;
    const std::string path = native_code_object_path();
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return fail_with(path + " is missing (build.py writes it)");
    std::vector<char> blob;
    { char buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) blob.insert(blob.end(), buf, buf + n); }
    fclose(f);
    hsa_code_object_reader_t reader;
    if (hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader) != HSA_STATUS_SUCCESS) return fail_with("code object reader");
    if (hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &nd->exe) != HSA_STATUS_SUCCESS) return fail_with("executable create");
    if (hsa_executable_load_agent_code_object(nd->exe, nd->agent, reader, nullptr, nullptr) != HSA_STATUS_SUCCESS) return fail_with("code object load (is " + path + " of this build?)");
    if (hsa_executable_freeze(nd->exe, nullptr) != HSA_STATUS_SUCCESS) return fail_with("executable freeze");
    // The code object must be the one this library was built with: a stale pair with another kernel-argument layout would dispatch garbage.  Both
    // translation units carry the build's source hash (build.py: -DMIRHI_SOURCE_HASH); the code object's copy is the device variable mirhi::g_build_id.
    {
        hsa_executable_symbol_t sym; uint64_t addr = 0; char theirs[32] = {0};
        if (hsa_executable_get_symbol_by_name(nd->exe, "_ZN5mirhi10g_build_idE", &nd->agent, &sym) != HSA_STATUS_SUCCESS ||
            hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_VARIABLE_ADDRESS, &addr) != HSA_STATUS_SUCCESS || !addr ||
            hsa_memory_copy(theirs, reinterpret_cast<void*>(addr), 17) != HSA_STATUS_SUCCESS)
            return fail_with(path + " carries no build id (built by another build.py?)");
        theirs[17] = 0;
        if (strcmp(theirs, MIRHI_SOURCE_HASH) != 0) return fail_with(path + " is of build " + theirs + ", this library of build " MIRHI_SOURCE_HASH);
    }
    nd->ok = true;
    return nd;
}

// the sRGB table of this copy of the code object (the HIP runtime's copy has its own: upload_srgb_lut)
bool native_upload_symbol(mirhi::NativeDevice* nd, const char* name, const void* src, size_t bytes) {
    hsa_executable_symbol_t sym;
    if (hsa_executable_get_symbol_by_name(nd->exe, name, &nd->agent, &sym) != HSA_STATUS_SUCCESS) return false;
    uint64_t addr = 0;
    if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_VARIABLE_ADDRESS, &addr) != HSA_STATUS_SUCCESS || !addr) return false;
    return hsa_memory_copy(reinterpret_cast<void*>(addr), src, bytes) == HSA_STATUS_SUCCESS;
}

// Is this queue the hardware's, or a software queue a tool put in front of it?  Under rocprofv3 counter collection (and any tool that registers with ROCr's
// queue-intercept API) hsa_queue_create returns a PROXY: its packets are consumed by a handler that runs inside the doorbell signal's store, rewritten
// (completion signals swapped, PM4 packets put around each dispatch) and copied to the real queue.  On a proxy `read_index` means "copied", not
// "consumed by the packet processor", so nothing about completion may be concluded from it.  The public signal ABI (hsa/amd_hsa_signal.h) tells the two apart:
// a hardware queue's doorbell signal is of kind AMD_SIGNAL_KIND_DOORBELL (a mapped doorbell register), a proxy's is an ordinary user signal.
bool native_queue_is_proxy(const hsa_queue_t* q) {
    const amd_signal_t* s = reinterpret_cast<const amd_signal_t*>(q->doorbell_signal.handle);
    return !s || (s->kind != AMD_SIGNAL_KIND_DOORBELL && s->kind != AMD_SIGNAL_KIND_LEGACY_DOORBELL);
}

mirhi::NativeQueue* native_queue_open(mirhi::NativeDevice* nd) {
    auto* nq = new mirhi::NativeQueue();
    nq->nd = nd;
    if (hsa_queue_create(nd->agent, NATIVE_QUEUE_PACKETS, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &nq->q) != HSA_STATUS_SUCCESS) { delete nq; return nullptr; }
    nq->proxy = native_queue_is_proxy(nq->q);
    nq->widx = hsa_queue_load_write_index_relaxed(nq->q);
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, NATIVE_RING_BYTES, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); (void)hsa_queue_destroy(nq->q); delete nq; return nullptr; }
    nq->ring = static_cast<uint8_t*>(p); nq->ring_bytes = NATIVE_RING_BYTES;
    if (hsa_signal_create(0, 0, nullptr, &nq->drain_sig) != HSA_STATUS_SUCCESS) { (void)hipFree(p); (void)hsa_queue_destroy(nq->q); delete nq; return nullptr; }
    return nq;
}
void native_queue_close(mirhi::NativeQueue* nq) {
    if (!nq) return;
    (void)hsa_signal_destroy(nq->drain_sig);
    (void)hsa_queue_destroy(nq->q);
    (void)hipFree(nq->ring);
    delete nq;
}

void native_mark_lost(mirhi::NativeDevice* nd, const char* where, const mirhi::NativeQueue* nq) {
    if (nd->lost.exchange(true)) return;
    char buf[320];
    if (nq) snprintf(buf, sizeof buf, "%s: no progress within %.1f s (queue write index %llu, read index %llu%s)", where, (double)nd->timeout_ns * 1e-9,
                     (unsigned long long)nq->widx, (unsigned long long)hsa_queue_load_read_index_relaxed(nq->q), nq->proxy ? ", intercepted queue" : "");
    else snprintf(buf, sizeof buf, "%s: no progress within %.1f s", where, (double)nd->timeout_ns * 1e-9);
    nd->lost_why = buf;
    fprintf(stderr, "mirhi: device lost: %s\n", buf);
}

// Waits until a completion signal reads 0.  Returns 0 done, 1 the caller's timeout expired, 2 device lost.  A frame's fence is due within
// microseconds: poll the word in host memory for 200 us, then block in the runtime (interrupt-driven, in slices of at most 10 ms so that a timeout
// is honoured) -- a thread that waits for seconds of GPU work must not keep a core spinning.  NO wait is unbounded: whatever the caller's timeout,
// when `watch`'s read index has not moved for the device's deadline (MIRHI_NATIVE_TIMEOUT_MS, 10 s) the device is marked lost -- a GPU that stopped
// consuming packets, or a tool between this library and the queue that does something else with them, becomes an error, never a hung thread.
int native_signal_wait(hsa_signal_t sig, uint64_t timeout_ns, mirhi::NativeDevice* nd, const mirhi::NativeQueue* watch, const char* where) {
    static const uint64_t ticks_per_s = [] { uint64_t f = 0; return hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &f) == HSA_STATUS_SUCCESS && f ? f : 100000000ull; }();
    if (nd && nd->lost.load(std::memory_order_acquire)) return hsa_signal_load_scacquire(sig) == 0 ? 0 : 2;
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t progress_at = 0, last_ridx = watch ? hsa_queue_load_read_index_relaxed(watch->q) : 0;
    for (uint32_t it = 0;; it++) {
        if (hsa_signal_load_scacquire(sig) == 0) return 0;
        cpu_relax();
        if ((it & 1023u) != 1023u) continue;
        const uint64_t elapsed = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        if (timeout_ns != UINT64_MAX && elapsed >= timeout_ns) return 1;
        if (watch) { const uint64_t r = hsa_queue_load_read_index_relaxed(watch->q); if (r != last_ridx) { last_ridx = r; progress_at = elapsed; } }
        if (nd && elapsed - progress_at >= nd->timeout_ns) { native_mark_lost(nd, where, watch); return 2; }
        if (elapsed >= 200000ull) {
            uint64_t slice_ns = 10000000ull;
            if (timeout_ns != UINT64_MAX && timeout_ns - elapsed < slice_ns) slice_ns = timeout_ns - elapsed;
            if (nd && nd->timeout_ns - (elapsed - progress_at) < slice_ns) slice_ns = nd->timeout_ns - (elapsed - progress_at);     // (never asleep past the deadline)
            (void)hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_EQ, 0, slice_ns * ticks_per_s / 1000000000ull + 1ull, HSA_WAIT_STATE_BLOCKED);
        }
    }
}

// Room for one more packet.  Kernel arguments: packet w uses slot w % SLOTS of the ring.  Every packet of this queue carries the barrier bit, so once the
// packet processor has CONSUMED packet j + 1 (read index >= j + 2), packet j has finished and its slot is free: at most SLOTS - 2 packets may be outstanding
// when slot w % SLOTS is written again.  On an intercepted queue the read index says "copied by the tool", nothing about the hardware (native_queue_is_proxy):
// there the queue is drained every SLOTS / 2 packets instead.  The wait is bounded: a read index that stands still for the device's deadline = device lost.
bool native_queue_drain_locked(mirhi::NativeQueue* nq);
bool native_reserve(mirhi::NativeQueue* nq) {
    mirhi::NativeDevice* nd = nq->nd;
    constexpr uint64_t SLOTS = NATIVE_RING_BYTES / NATIVE_KERNARG_SLOT;
    static_assert(SLOTS >= 64 && SLOTS <= NATIVE_QUEUE_PACKETS, "kernarg ring");
    if (nd->lost.load(std::memory_order_acquire)) return false;
    if (nq->proxy) { if (nq->widx - nq->drained >= SLOTS / 2) return native_queue_drain_locked(nq); return true; }
    if (nq->widx - hsa_queue_load_read_index_scacquire(nq->q) <= SLOTS - 2) return true;
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t last = hsa_queue_load_read_index_relaxed(nq->q), progress_ns = 0;
    for (uint32_t spins = 1;; spins++) {          // (back-pressure of a full queue: the producer yields)
        const uint64_t r = hsa_queue_load_read_index_scacquire(nq->q);
        if (nq->widx - r <= SLOTS - 2) return true;
        cpu_relax();
        if ((spins & 4095u) != 0u) continue;
        std::this_thread::yield();
        const uint64_t elapsed = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        if (r != last) { last = r; progress_ns = elapsed; }
        if (elapsed - progress_ns >= nd->timeout_ns) { native_mark_lost(nd, "waiting for room in an AQL queue", nq); return false; }
    }
}
// Publishes the packet at index nq->widx whose body has been written: header (release), write index, doorbell -- the order ROCr documents for a
// single-producer queue (the index was reserved by this one producer: nq->mu).
void native_publish(mirhi::NativeQueue* nq, void* packet, uint16_t header, uint16_t setup) {
    __atomic_store_n(reinterpret_cast<uint32_t*>(packet), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);     // (also orders the kernarg stores: x86 stores stay in order)
    const uint64_t idx = nq->widx++;
    hsa_queue_store_write_index_screlease(nq->q, idx + 1);
    __builtin_ia32_sfence();                                     // the write-combined kernarg stores leave the CPU before the doorbell does
    hsa_signal_store_screlease(nq->q->doorbell_signal, (hsa_signal_value_t)idx);
}

// every packet written to the queue so far has completed (a barrier packet with a completion signal, waited for on the host); false: device lost
bool native_queue_drain_locked(mirhi::NativeQueue* nq) {
    if (nq->drained == nq->widx) return !nq->nd->lost.load(std::memory_order_acquire);
    if (nq->nd->lost.load(std::memory_order_acquire)) return false;
    // (ring room: NATIVE_QUEUE_PACKETS is four times the packets native_reserve lets be outstanding)
    hsa_signal_store_relaxed(nq->drain_sig, 1);
    auto* p = reinterpret_cast<hsa_barrier_and_packet_t*>(nq->q->base_address) + (nq->widx & (NATIVE_QUEUE_PACKETS - 1));
    memset(reinterpret_cast<uint8_t*>(p) + 4, 0, sizeof *p - 4);
    p->completion_signal = nq->drain_sig;
    const uint16_t header = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    native_publish(nq, p, header, 0);
    if (native_signal_wait(nq->drain_sig, UINT64_MAX, nq->nd, nq, "draining an AQL queue") != 0) return false;
    nq->drained = nq->widx;
    return true;
}
bool native_queue_drain(mirhi::NativeQueue* nq) {
    if (!nq) return true;
    std::lock_guard<std::mutex> qlock(nq->mu);
    return native_queue_drain_locked(nq);
}
}  // namespace

hipError_t mirhi::native_enqueue(NativeQueue* nq, const void* key, dim3 grid, dim3 block, const void* args, size_t args_bytes, uint64_t signal, uint32_t flags) {
    NativeDevice* nd = nq->nd;
    const NativeKernel* k = nullptr;
    {
        std::lock_guard<std::mutex> lock(nd->mu);
        for (auto& e : nd->kernels) if (e.first == key) { k = &e.second; break; }
        if (!k) {
            hsa_executable_symbol_t sym;
            const char* mangled = hipKernelNameRefByPtr(key, nullptr);         // the device-side symbol the HIP runtime registered for this kernel
            if (!mangled) { (void)hipGetLastError(); return hipErrorInvalidDeviceFunction; }
            const std::string name = std::string(mangled) + ".kd";
            NativeKernel nk{};
            if (hsa_executable_get_symbol_by_name(nd->exe, name.c_str(), &nd->agent, &sym) != HSA_STATUS_SUCCESS ||
                hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &nk.object) != HSA_STATUS_SUCCESS ||
                hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &nk.kernarg_bytes) != HSA_STATUS_SUCCESS ||
                hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &nk.lds_bytes) != HSA_STATUS_SUCCESS ||
                hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &nk.scratch_bytes) != HSA_STATUS_SUCCESS)
                return hipErrorInvalidDeviceFunction;
            if (nk.scratch_bytes != 0) return hipErrorInvalidDeviceFunction;         // (no kernel of this library uses scratch; a queue of ours has none set up)
            nd->kernels.emplace_back(key, nk);
            k = &nd->kernels.back().second;
        }
    }
    std::lock_guard<std::mutex> qlock(nq->mu);
    const size_t explicit_bytes = (args_bytes + 7) & ~(size_t)7;
    const size_t need = (std::max<size_t>(explicit_bytes + sizeof(ImplicitArgs), k->kernarg_bytes) + 63) & ~(size_t)63;
    if (need > NATIVE_KERNARG_SLOT || explicit_bytes + sizeof(ImplicitArgs) > NATIVE_KERNARG_SLOT) return hipErrorInvalidValue;
    if (!native_reserve(nq)) return hipErrorLaunchTimeOut;          // (device lost: reported as such by the caller)
    constexpr uint64_t SLOTS = NATIVE_RING_BYTES / NATIVE_KERNARG_SLOT;
    uint8_t* ka = nq->ring + (nq->widx % SLOTS) * NATIVE_KERNARG_SLOT;
    alignas(16) uint8_t tmp[NATIVE_KERNARG_SLOT];
    memcpy(tmp, args, args_bytes);
    if (explicit_bytes > args_bytes) memset(tmp + args_bytes, 0, explicit_bytes - args_bytes);
    ImplicitArgs im;
    memset(&im, 0, sizeof im);
    im.block_count[0] = grid.x; im.block_count[1] = grid.y; im.block_count[2] = grid.z;
    im.group_size[0] = (uint16_t)block.x; im.group_size[1] = (uint16_t)block.y; im.group_size[2] = (uint16_t)block.z;
    im.grid_dims = grid.z > 1 ? 3 : (grid.y > 1 ? 2 : 1);
    memcpy(tmp + explicit_bytes, &im, sizeof im);
    memcpy(ka, tmp, explicit_bytes + sizeof im);             // write-combined stores over the BAR
    auto* p = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(nq->q->base_address) + (nq->widx & (NATIVE_QUEUE_PACKETS - 1));
    p->workgroup_size_x = (uint16_t)block.x; p->workgroup_size_y = (uint16_t)block.y; p->workgroup_size_z = (uint16_t)block.z;
    p->reserved0 = 0;
    p->grid_size_x = grid.x * block.x; p->grid_size_y = grid.y * block.y; p->grid_size_z = grid.z * block.z;
    p->private_segment_size = 0; p->group_segment_size = k->lds_bytes;
    p->kernel_object = k->object; p->kernarg_address = ka; p->reserved2 = 0;
    p->completion_signal.handle = signal;
    // every kernel waits for the one before it on its queue (barrier bit); its fences are agent scope unless the caller asked for system scope
    // (a system-scope ACQUIRE costs the packet 2.6 us -- tools/microbench/hsa/hsa_dispatch.cpp -- so it is asked for only when something
    // outside this agent wrote memory since the queue's last one; a completion signal asks for a system-scope RELEASE: the host is told)
    const uint16_t acq = (flags & NATIVE_ACQUIRE_SYSTEM) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
    const uint16_t rel = (flags & NATIVE_RELEASE_SYSTEM) || signal ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            (uint16_t)(acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (uint16_t)(rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    nd->dispatches.fetch_add(1, std::memory_order_relaxed);
    native_publish(nq, p, header, 3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS);
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// objects
// ------------------------------------------------------------------------------------------------
// One timed dispatch: the event pair is attached to the dispatch itself (hipExtLaunchKernelGGL), so elapsed(start, stop) is
// the kernel's begin -> end on the GPU clock and elapsed(stop of A, stop of B) the distance between two kernels' ends.
struct TimedDispatch { hipEvent_t start, stop; uint32_t kernel, lane; };

struct mirhi_device {
    int ordinal = 0;
    hipStream_t stream = nullptr;           // lane 0
    bool owns_stream = false;
    std::vector<hipStream_t> lanes;          // submit streams; lanes[0] == stream
    uint32_t next_lane = 0;
    std::atomic<int> children{0};
    // Bumped whenever something that is not a kernel of this library writes device memory the kernels may read (copy engines, HIP's
    // memsets, RCCL, memory wrapped from outside): the next scope on each AQL queue then opens with a system-scope acquire; all others
    // open with an agent-scope one (tools/microbench/hsa/stale.cpp: no stale word after any such write even without -- kept as the rule)
    std::atomic<uint64_t> foreign_writes{1};
    uint32_t split_rank = 0, split_world = 1;
    uint32_t split_layout = MIRHI_SPLIT_INTERLEAVED;   // mirhi_device_set_tile_split_layout (MIRHI_SPLIT=bands|interleaved sets the default)
    uint32_t profiling = 0;                  // MIRHI_PROFILE_* bits
    std::mutex mu;
    std::vector<TimedDispatch> pending;       // timed dispatches not yet read back
    std::vector<mirhi_dispatch_time> timeline;   // read back: begin / end relative to the first one since the last reset
    hipEvent_t base_stop = nullptr;           // stop event of that first dispatch (kept until the next reset)
    double base_end_ms = 0.0;                 //   its end on the timeline (= its own duration: the timeline starts at its begin)
    std::vector<hipEvent_t> free_events;
    std::vector<mirhi_cmd*> unchecked;        // submitted since the last wait_idle: their status words are read there
    std::vector<mirhi_fence*> fences;         // live fences (guarded by mu): a command buffer that is destroyed or re-recorded
                                              //   while a fence still lists it hands its device status over first
    std::vector<mirhi_cmd*> cmds;             // live command buffers (guarded by mu): lanes that go away are forgotten by all of them
    std::vector<mirhi_image*> images;         // live images (guarded by mu): a recording may outlive an attachment it names
    hipEvent_t order_event = nullptr;         // cross-lane attachment ordering (mirhi_image::last_stream)
    // native dispatch (mirhi_native.h): one AQL queue per queue lane, opened when the lane first carries a native submit
    NativeDevice* native = nullptr;
    std::vector<NativeQueue*> native_lanes;
    std::vector<hipStream_t> aux_streams;     // streams of the device's communicators (band exchange): wait_idle waits for them, an image they wrote last is ordered behind them
    bool native_on_external = false;          // mirhi_device_set_native_dispatch: lane 0 of a device made on the caller's stream dispatches natively too
    mirhi_result deferred = MIRHI_OK;         // status of such a command buffer that no fence listed: reported by wait_idle
    std::string deferred_msg;
    double total_ms[MIRHI_KERNEL_COUNT] = {0, 0, 0, 0};
    uint64_t launches[MIRHI_KERNEL_COUNT] = {0, 0, 0, 0};
    unsigned long long* frag_stats = nullptr; // device: [0] pixels that ran a fragment program, [1] covered fragments (profiling only)
    uint64_t frag_scopes = 0;                 // scopes counted into frag_stats since the last reset
    mirhi_device_stats stats{};
    char name[256] = {0};
    // Submit thread (mirhi_device_set_submit_thread): vkQueueSubmit hands the work to the driver and returns; with the thread on,
    // mirhi_queue_submit validates, queues the job and returns, and this thread makes the HIP launches (2.3 - 3 us each whatever the
    // entry point, tools/microbench/launch_paths.hip: 5 - 7 us per frame that the render thread then spends recording the next frame).
    // Single producer / single consumer ring; the consumer spins while a frame loop feeds it and sleeps after 100 us without work.
    struct SubmitJob { uint32_t n; mirhi_cmd* cmds[MAX_BATCH > 16 ? MAX_BATCH : 16]; mirhi_fence* fence; };
    static constexpr uint64_t SQ_SLOTS = 64;
    std::thread sq_thread;
    bool sq_on = false;
    SubmitJob sq_ring[SQ_SLOTS];
    std::atomic<uint64_t> sq_pushed{0}, sq_done{0};
    std::atomic<bool> sq_stop{false}, sq_sleeping{false};
    std::mutex sq_mu; std::condition_variable sq_cv;
    std::mutex sq_push_mu;                                    // producers: several host threads may submit to one device (the ring has ONE consumer)
};

struct mirhi_buffer {
    mirhi_device* dev;
    mirhi_buffer_usage usage;
    uint64_t size;
    uint8_t* ptr;
    bool owned;
    bool host_direct = false;   // fine-grained device memory the host stores into (uniform buffers): write_data is a memcpy, as in the reference
};

struct mirhi_image {
    mirhi_device* dev;
    uint32_t width, height;
    mirhi_format format;
    uint8_t* ptr;
    bool owned;
    uint32_t levels = 1;      // mip levels stored contiguously behind level 0 (mirhi_image_generate_mips)
    uint32_t max_anisotropy = 1;   // sampler state (mirhi_image_set_max_anisotropy): 1 = trilinear
    // Attachment ordering across queue lanes: the reference submits everything to one queue, so a scope that LOADs (or overwrites)
    // what an earlier submission rendered is behind it by construction; here two command buffers may sit on different lanes.  The
    // stream, command buffer and submission that used this image as an attachment last: a submit on ANOTHER stream waits for it
    // (one event record + stream wait), unless a fence / wait_idle has reported that submission finished -- the frame loop's case.
    hipStream_t last_stream = nullptr; const mirhi_cmd* last_cmd = nullptr; uint64_t last_seq = 0;
    NativeQueue* last_native = nullptr;      // ... or the AQL queue, when that submission was dispatched natively
};

struct mirhi_pipeline {
    mirhi_device* dev;
    mirhi_pipeline_desc desc;
};

struct RecordedPass {
    mirhi_rendering_info info;
    // what the attachments resolved to when the scope was recorded: two recordings are the same frame shape only if these agree
    // (an image handle may be a new image at an old address)
    struct Target { const uint8_t* ptr; uint32_t width, height, format; };
    Target color_t{nullptr, 0, 0, 0}, depth_t{nullptr, 0, 0, 0}, prim_t{nullptr, 0, 0, 0};
    std::vector<DrawDesc> draws;
    std::vector<uint64_t> draw_vb_bytes;   // bytes of the bound vertex buffer range per draw (vertex pre-pass extent)
    uint32_t total_tris = 0;
    bool key_set = false;
    uint32_t depth_test = 0, depth_compare = 0, depth_write = 0, frag_discard = 0;
    uint32_t blend[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // enable, src colour, dst colour, colour op, src alpha, dst alpha, alpha op, write mask
    // a scope whose pipelines change the depth state is continued in a new segment: colour is kept (LOAD), depth is
    // carried through the depth attachment or, without one, a transient buffer of the workspace
    uint32_t first_tri = 0;                // primitive ids continue across the segments of a scope
    bool carry_in = false, carry_out = false;
    int32_t area[4] = {0, 0, 0, 0};
};

// Words of the counter block that never move (a re-recorded frame of another shape finds them where the last frame's kernels
// left them, re-armed): the two big-list counters, the eight pool counters a cache line apart; the bin counters follow.
constexpr uint32_t CTR_BIG = 0u, CTR_POOL = 32u, CTR_ACTIVE = CTR_POOL + 8u * (uint32_t)POOL_COUNTER_STRIDE,   // busy-tile counters: two parities x 8 x 32 words
                   CTR_BINS = CTR_ACTIVE + 2u * 8u * 32u + 32u;

struct Workspace {
    // Parameter block: everything the kernels read as "parameters" -- two PassParams per scope (big-list counter parity 0 / 1), the
    // draw descriptors, the vertex pre-pass jobs -- in ONE piece of fine-grained device memory that the host writes directly
    // (stores over the PCIe BAR, 0.2 us per KB; tools/microbench/fence_latency.hip) and only when its bytes changed: re-recording
    // an unchanged frame uploads nothing, a changed one costs a memcpy, and no copy command ever enters a stream.  `pshadow` is the
    // host's copy of what the block holds (device memory is never read back over the BAR).
    uint8_t* pblock = nullptr; size_t pblock_bytes = 0; bool pblock_direct = false;   // direct: host-writable; else uploads go through `pstage`
    bool foreign = false;                                                             // the block went up through a copy engine (fallback): the next scope acquires at system scope
    uint8_t* pstage = nullptr; size_t pstage_bytes = 0;                                // pinned staging for the fallback (hipMemcpyAsync on the lane)
    std::vector<uint8_t> pshadow, pimage;                                              // what the block holds / what it should hold
    size_t draws_off = 0, jobs_off = 0, draws_count = 0;
    bool dirty = true;                                         // counters / page table not known to be in their idle state: cleared at the next plan
    BinRec* bin_pool = nullptr; size_t bin_pool_bytes = 0;     // pages of BIN_PAGE_RECS records: fixed first pages, then the dynamic ones
    uint32_t* bin_table = nullptr; size_t bin_table_bytes = 0; // [tiles][BIN_TABLE_ROW] page table, PAGE_EMPTY when idle
    uint32_t pool_scale = 1;                                   // doubled whenever a scope exhausted the pool (applied at the next submit)
    bool grow_pool = false;
    // Feedback on the two-team mesh mode (raster_mode picks it from the triangle count alone: 'few triangles per tile' is a mesh
    // that sits in a part of the frame -- or one spread thinly over all of it, where the wider workgroups only cost occupancy): a
    // scope that ran in that mode and opened more than two list pages per tile of the frame was spread out (64x64-quad grid over a
    // 1080p frame: 5,340 pages on 2,040 tiles, raster 41.8 us with two teams, 30.4 us with one; the dancer asset: 1,770 pages, 39 us
    // against 66 us); the plan is rebuilt with one team.
    bool spread = false, replan = false;
    // Feedback on the wide mesh variant (sixteen waves per tile: raster_body, WIDE): the busy tiles the scope before last reported.  A mesh
    // scope whose triangles sit in at most WIDE_ON tiles leaves most of the chip without work -- every busy tile then gets four times the
    // waves; above WIDE_OFF it goes back (hysteresis: a frame loop must not flip between two plans).
    uint32_t busy_tiles = 0; bool busy_known = false, wide_eligible = false;
    uint32_t wide = 0;                                         // waves per tile of the wide variant in use: 0 (four waves), 8 or 16
    uint64_t spread_tris = 0;                                  // triangles of the command buffer when `spread` was measured: forgotten when the
                                                               // recorded frame is another one (more than twice / less than half as many)
    uint32_t xcd_tiles_last = 0;                               // tiles of the command buffer's last scope if it uses per-XCD bins, else 0
    uint32_t* counters = nullptr; size_t counters_words = 0;   // [8 * tiles] bin counts, two big-list counters, the pool counter
    BigRec* big_recs = nullptr; size_t big_recs_bytes = 0;
    float* carry_depth = nullptr; size_t carry_depth_bytes = 0;   // depth hand-over between the segments of a scope without a depth attachment
    TriRec* ordered = nullptr; size_t ordered_bytes = 0;          // slot t = triangle t of an ordered segment (blending)
    PassParams* params = nullptr;                            // = pblock: two copies per scope (big-list counter parity 0 / 1), read by the kernels
    uint8_t* vs_out = nullptr; size_t vs_out_bytes = 0;
    uint32_t* flat_color = nullptr; size_t flat_color_bytes = 0;
    uint32_t* prim_draw = nullptr; size_t prim_draw_bytes = 0;     // per primitive: its draw (scopes with several draws)
    uint32_t* status_host = nullptr;                             // pinned, device-mapped: [status bits, big-list length]
    uint32_t* status_dev = nullptr;                              // device view of status_host
    uint32_t* big_counts = nullptr;                              // two counters, used alternately (parity)
    uint32_t parity = 0;
    // statistics pass (MIRHI_PROFILE_FRAGMENTS), built on first use: a primitive-id image of the workspace's own and copies
    // of the scopes' parameters that write it, so that the product kernels and parameters stay untouched
    uint32_t* stats_prim = nullptr; size_t stats_prim_bytes = 0;
    PassParams* stats_params = nullptr; size_t stats_params_bytes = 0; bool stats_params_valid = false;
    size_t bytes() const { return pblock_bytes + bin_pool_bytes + bin_table_bytes + counters_words * 4 + big_recs_bytes + vs_out_bytes + flat_color_bytes; }
};

enum CmdState { CMD_INITIAL = 0, CMD_RECORDING = 1, CMD_EXECUTABLE = 2 };

struct mirhi_cmd {
    mirhi_device* dev;
    CmdState state = CMD_INITIAL;
    uint32_t lane = 0;                     // submit stream of this command buffer (frames in flight overlap across lanes)
    hipStream_t last_stream = nullptr;     // the stream its last submission ran on (a batched submit runs on the first command buffer's lane)
    bool pending = false;                  // submitted and not yet known to have finished (a fence wait / wait_idle clears it).  Vulkan forbids
                                           //   re-recording a pending command buffer; this build synchronises instead -- the slow path a frame
                                           //   loop that waits on its fences never takes
    uint64_t submit_seq = 0;               // submissions so far (a fence remembers which one it saw)
    std::atomic<int> queued{0};            // submissions handed to the submit thread and not yet issued by it
    NativeQueue* last_native = nullptr;    // its last submission went out as AQL packets on this queue (else: last_stream)
    bool one_time = true;
    bool in_rendering = false;
    std::vector<RecordedPass> passes;
    // Plan cache: the recording the current plan was built from.  A frame loop records the same frame again and again
    // (renderer.rs:452-557); end() then finds plan, workspace and parameter block as they are and touches nothing.
    std::vector<RecordedPass> planned;
    bool plan_valid = false;
    uint32_t plan_split_rank = 0, plan_split_world = 1, plan_split_layout = 0;
    // current bindings (dynamic state + descriptors)
    mirhi_pipeline* pipeline = nullptr;
    mirhi_buffer* vb = nullptr; uint64_t vb_offset = 0;
    mirhi_buffer* ib = nullptr; uint64_t ib_offset = 0; mirhi_index_type ib_type = MIRHI_INDEX_UINT32;
    struct { mirhi_buffer* buf; uint64_t offset, range; } uniforms[MIRHI_SLOT_COUNT] = {};
    mirhi_image* textures[MIRHI_TEXTURE_COUNT] = {};
    bool has_viewport = false, has_scissor = false;
    uint8_t push_constants[128] = {0};     // vkCmdPushConstants: kept, read by no program on this path
    mirhi_viewport viewport{};
    mirhi_rect2d scissor{};
    // device-side plan, built at end()
    Workspace ws;
    std::vector<PassParams> plan;
    std::vector<uint32_t> plan_programs;
    uint64_t plan_tris = 0;
};

struct mirhi_fence {
    mirhi_device* dev;
    hipEvent_t event = nullptr;
    hipEvent_t join = nullptr;   // cross-lane join for multi-command submits
    bool signaled = false;      // host-visible signaled state
    bool pending = false;       // an event record is outstanding
    std::atomic<bool> issued{true};   // false while its submission waits in the submit thread's queue (the event is recorded when it is issued)
    hsa_signal_t native_sig{0};       // native dispatch: completion signal of the submission's last packet (1 -> 0), polled in host memory
    bool native_wait = false;         //   the pending submission is waited for through it, not through `event`
    NativeQueue* native_q = nullptr;  //   the queue that carries it (its read index is the progress a bounded wait watches)
    std::vector<mirhi_cmd*> cmds;  // submissions to check for device status on completion
    std::vector<uint64_t> seqs;    //   and which submission of each it was (mirhi_cmd::submit_seq)
    mirhi_result deferred = MIRHI_OK;   // status handed over by a listed command buffer that was destroyed / re-recorded since
    std::string deferred_msg;
};

// the command buffer's last submission is known to have finished
static inline void cmd_finished(mirhi_cmd* c) { c->pending = false; }

static uint32_t format_bpp(mirhi_format f) {
    switch (f) {
        case MIRHI_FORMAT_B8G8R8A8_SRGB: case MIRHI_FORMAT_D32_SFLOAT: case MIRHI_FORMAT_R8G8B8A8_UNORM:
        case MIRHI_FORMAT_R8G8B8A8_SRGB: case MIRHI_FORMAT_R32_UINT: return 4;
        case MIRHI_FORMAT_R32G32B32A32_SFLOAT: return 16;
        default: return 0;
    }
}

// ------------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------------
extern "C" mirhi_result mirhi_device_count(int32_t* out_count) {
    NULL_CHECK(out_count, "out_count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out_count = 0; (void)hipGetLastError(); return MIRHI_OK; }
    *out_count = n;
    return MIRHI_OK;
}

static mirhi_result device_create_common(int32_t ordinal, void* stream, bool external, mirhi_device** out) {
    NULL_CHECK(out, "out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { (void)hipGetLastError(); return fail(MIRHI_ERR_NO_SUITABLE_GPU, "No suitable GPU found"); }
    if (ordinal < 0 || ordinal >= n) return fail(MIRHI_ERR_NO_SUITABLE_GPU, "No suitable GPU found (ordinal %d of %d)", ordinal, n);
    HIP_TRY(hipSetDevice(ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MIRHI_ERR_NO_SUITABLE_GPU, "No suitable GPU found (device %d is %s, kernels are built for gfx950)", ordinal, prop.gcnArchName);
    mirhi_device* d = new (std::nothrow) mirhi_device();
    if (!d) return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: host allocation failed");
    d->ordinal = ordinal;
    if (const char* sp = getenv("MIRHI_SPLIT")) d->split_layout = strcmp(sp, "bands") == 0 ? (uint32_t)MIRHI_SPLIT_BANDS : (uint32_t)MIRHI_SPLIT_INTERLEAVED;
    snprintf(d->name, sizeof d->name, "%s (%s)", prop.name, prop.gcnArchName);
    if (external) { d->stream = (hipStream_t)stream; d->owns_stream = false; }
    else {
        hipError_t se = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
        if (se != hipSuccess) { delete d; return hip_fail(se, "hipStreamCreateWithFlags"); }
        d->owns_stream = true;
    }
    d->lanes.push_back(d->stream);
    {
        const hipError_t fe = hipMalloc((void**)&d->frag_stats, 2 * sizeof(unsigned long long));
        // (nothing here goes through the NULL stream: it would claim one of the process's few hardware queues -- ROCclr keeps
        // GPU_MAX_HW_QUEUES = 4 and lets further streams share them -- and two queue lanes that share a hardware queue run their
        // frames one after the other: C2 12.7 us per frame instead of 8.0, tools/saturation_probe.py)
        if (fe != hipSuccess || hipMemsetAsync(d->frag_stats, 0, 2 * sizeof(unsigned long long), d->stream) != hipSuccess) {
            if (d->owns_stream) (void)hipStreamDestroy(d->stream);
            delete d; (void)hipGetLastError();
            return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: hipMalloc for the device statistics block");
        }
    }
    {   // sRGB EOTF per byte, evaluated in double and rounded once (the oracle builds the identical table)
        float lut[256];
        for (int i = 0; i < 256; i++) {
            const double c = (double)i / 255.0;
            lut[i] = (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
        }
        hipError_t le = upload_srgb_lut(lut, d->stream);
        if (le == hipSuccess) le = hipStreamSynchronize(d->stream);
        if (le != hipSuccess) { if (d->owns_stream) (void)hipStreamDestroy(d->stream); delete d; return hip_fail(le, "sRGB table upload"); }
        // native dispatch: the same kernels through our own AQL queues (mirhi_native.h); its copy of the code object has its own table
        d->native = native_device_open(ordinal);
        if (d->native->ok && !native_upload_symbol(d->native, "_ZN5mirhi10g_srgb_lutE", lut, sizeof lut)) { d->native->ok = false; d->native->why = "sRGB table symbol not found in the code object"; }
        d->native_lanes.assign(1, nullptr);
        if (!d->native->ok && native_env().dispatch == 2) {      // (2: required -- tests that must not pass on the fallback)
            const std::string why = d->native->why;
            if (d->owns_stream) (void)hipStreamDestroy(d->stream);
            delete d->native; delete d;
            return fail(MIRHI_ERR_LOADING, "Loading error: native dispatch unavailable: %s", why.c_str());
        }
    }
    *out = d;
    if (getenv("MIRHI_SUBMIT_THREAD") && atoi(getenv("MIRHI_SUBMIT_THREAD")) != 0) (void)mirhi_device_set_submit_thread(d, 1);   // (test runs: every device with the thread on)
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_create(int32_t ordinal, mirhi_device** out) { return device_create_common(ordinal, nullptr, false, out); }
extern "C" mirhi_result mirhi_device_create_on_stream(int32_t ordinal, void* stream, mirhi_device** out) { return device_create_common(ordinal, stream, true, out); }

// every job handed to the submit thread has been issued to the GPU's queues (not: has finished)
static void drain_submits(mirhi_device* dev) {
    if (!dev->sq_on) return;
    const uint64_t want = dev->sq_pushed.load(std::memory_order_acquire);
    while (dev->sq_done.load(std::memory_order_acquire) < want) cpu_relax();
}

// A native wait ran into its deadline (native_mark_lost): the device is gone as far as this library can tell -- VK_ERROR_DEVICE_LOST
static mirhi_result device_lost(mirhi_device* dev) {
    return fail(MIRHI_ERR_DEVICE, "Vulkan error: DEVICE_LOST: %s", dev->native ? dev->native->lost_why.c_str() : "native dispatch");
}
static bool drain_native(mirhi_device* dev, const NativeQueue* which) {      // drains `which` if it is (still) one of the device's queues; false: device lost
    for (NativeQueue* q : dev->native_lanes) if (q && q == which) return native_queue_drain(q);
    return true;
}
static mirhi_result sync_all_lanes(mirhi_device* dev) {
    drain_submits(dev);
    HIP_TRY(hipSetDevice(dev->ordinal));
    bool ok = true;
    for (NativeQueue* nq : dev->native_lanes) ok = native_queue_drain(nq) && ok;
    for (hipStream_t st : dev->lanes) HIP_TRY(hipStreamSynchronize(st));
    for (hipStream_t st : dev->aux_streams) HIP_TRY(hipStreamSynchronize(st));
    return ok ? MIRHI_OK : device_lost(dev);
}
// the AQL queue of a queue lane (opened on first use; native_lanes is sized with the lanes -- mirhi_device_set_queue_lanes -- so that threads that walk it
// never see it move); nullptr: native dispatch is not available
static NativeQueue* native_lane(mirhi_device* dev, uint32_t lane) {
    if (!dev->native || !dev->native->ok || dev->native->lost.load(std::memory_order_acquire) || lane >= dev->native_lanes.size()) return nullptr;
    if (!dev->native_lanes[lane]) {
        NativeQueue* nq = native_queue_open(dev->native);
        if (!nq) { dev->native->ok = false; dev->native->why = "could not open an AQL queue"; return nullptr; }
        if (nq->proxy && native_env().dispatch != 2) {
            // A tool sits between this library and the hardware queue (rocprofv3 --pmc, anything on ROCr's queue-intercept API): its packets would be
            // rewritten and the read index would stop meaning what the producer relies on.  The launches go through HIP, whose queues the tool expects.
            native_queue_close(nq);
            dev->native->ok = false; dev->native->why = "the AQL queue is intercepted by a tool (software doorbell): launches go through HIP";
            return nullptr;
        }
        dev->native_lanes[lane] = nq;
    }
    return dev->native_lanes[lane];
}
static mirhi_result check_status_words(mirhi_device* dev);
extern "C" mirhi_result mirhi_device_set_submit_thread(mirhi_device* dev, uint32_t enable);
extern "C" mirhi_result mirhi_device_wait_idle(mirhi_device* dev) {
    NULL_CHECK(dev, "device");
    mirhi_result r = sync_all_lanes(dev);
    if (r != MIRHI_OK) return r;
    return check_status_words(dev);         // a frame that went wrong on the device fails here too, not only at a fence
}
extern "C" mirhi_result mirhi_device_set_queue_lanes(mirhi_device* dev, uint32_t lanes) {
    NULL_CHECK(dev, "device");
    if (lanes < 1 || lanes > 8) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: queue lanes must be 1..8 (got %u)", lanes);
    mirhi_result r = sync_all_lanes(dev);
    if (r != MIRHI_OK) return r;
    if (dev->lanes.size() > lanes) {
        // (a command buffer must not keep the handle of a stream that is about to go: ADVICE r2, mirhi_cmd::last_stream)
        std::lock_guard<std::mutex> lock(dev->mu);
        for (mirhi_cmd* c : dev->cmds) { c->last_stream = nullptr; c->last_native = nullptr; cmd_finished(c); }
        for (mirhi_image* img : dev->images) { img->last_stream = nullptr; img->last_native = nullptr; img->last_cmd = nullptr; }
    }
    while (dev->lanes.size() > lanes) { (void)hipStreamDestroy(dev->lanes.back()); dev->lanes.pop_back(); }
    {
        std::lock_guard<std::mutex> lock(dev->mu);
        while (dev->native_lanes.size() > lanes) { native_queue_close(dev->native_lanes.back()); dev->native_lanes.pop_back(); }
        dev->native_lanes.resize(lanes, nullptr);
    }
    while (dev->lanes.size() < lanes) {
        hipStream_t st = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        dev->lanes.push_back(st);
    }
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_destroy(mirhi_device* dev) {
    NULL_CHECK(dev, "device");
    if (dev->children.load() != 0)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: device still has %d live child objects", dev->children.load());
    (void)mirhi_device_set_submit_thread(dev, 0);
    (void)sync_all_lanes(dev);
    for (size_t i = 1; i < dev->lanes.size(); i++) (void)hipStreamDestroy(dev->lanes[i]);
    for (auto& p : dev->pending) { if (p.start != dev->base_stop) (void)hipEventDestroy(p.start); if (p.stop != dev->base_stop) (void)hipEventDestroy(p.stop); }
    if (dev->base_stop) (void)hipEventDestroy(dev->base_stop);
    for (auto& e : dev->free_events) (void)hipEventDestroy(e);
    if (dev->frag_stats) (void)hipFree(dev->frag_stats);
    if (dev->order_event) (void)hipEventDestroy(dev->order_event);
    for (NativeQueue* nq : dev->native_lanes) native_queue_close(nq);
    if (dev->native) {
        if (dev->native->exe.handle) (void)hsa_executable_destroy(dev->native->exe);
        delete dev->native;
    }
    if (dev->owns_stream) (void)hipStreamDestroy(dev->stream);
    delete dev;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_name(mirhi_device* dev, char* out, uint32_t out_len) {
    NULL_CHECK(dev, "device"); NULL_CHECK(out, "out");
    if (out_len == 0) return MIRHI_OK;
    snprintf(out, out_len, "%s", dev->name);
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_set_tile_split(mirhi_device* dev, uint32_t rank, uint32_t world) {
    NULL_CHECK(dev, "device");
    if (world == 0 || rank >= world) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: tile split rank %u of %u", rank, world);
    drain_submits(dev);
    dev->split_rank = rank; dev->split_world = world;
    return MIRHI_OK;
}
// The tile rows of a frame of `tiles_y` rows that `rank` of `world` rasterizes: rows first + k * step, k < count (PassParams::tile_row_*).
//   bands        one contiguous band per rank, ceil(tiles_y / world) rows, the last rank short: step 1
//   interleaved  rank r owns rows r, r + world, r + 2 world, ...: every rank gets the same share of every part of the frame, whatever the scene puts where
//                (SURVEY 8e's load-balance option; C5 in eight bands: 71 | 72 | 70 | 66 | 53 | 24 | 11 | 5 us of raster -- the slowest rank sets the frame)
static void split_rows(uint32_t layout, uint32_t rank, uint32_t world, uint32_t tiles_y, uint32_t* first, uint32_t* step, uint32_t* count) {
    if (world <= 1) { *first = 0; *step = 1; *count = tiles_y; return; }
    if (layout == MIRHI_SPLIT_BANDS) {
        const uint32_t per = (tiles_y + world - 1) / world;
        uint32_t b = rank * per, e = b + per;
        if (b > tiles_y) b = tiles_y;
        if (e > tiles_y) e = tiles_y;
        *first = b; *step = 1; *count = e - b;
        return;
    }
    *first = rank; *step = world; *count = rank < tiles_y ? (tiles_y - rank + world - 1) / world : 0u;
}
extern "C" mirhi_result mirhi_device_set_tile_split_layout(mirhi_device* dev, mirhi_split_layout layout) {
    NULL_CHECK(dev, "device");
    if (layout != MIRHI_SPLIT_BANDS && layout != MIRHI_SPLIT_INTERLEAVED) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: unknown tile split layout %d", (int)layout);
    drain_submits(dev);
    dev->split_layout = (uint32_t)layout;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_split_rows(mirhi_device* dev, uint32_t height, uint32_t* first_tile_row, uint32_t* tile_row_step, uint32_t* tile_rows) {
    NULL_CHECK(dev, "device"); NULL_CHECK(first_tile_row, "first_tile_row"); NULL_CHECK(tile_row_step, "tile_row_step"); NULL_CHECK(tile_rows, "tile_rows");
    split_rows(dev->split_layout, dev->split_rank, dev->split_world, (height + TILE - 1) / TILE, first_tile_row, tile_row_step, tile_rows);
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_band_rows(mirhi_device* dev, uint32_t height, uint32_t* row_begin, uint32_t* row_end) {
    NULL_CHECK(dev, "device"); NULL_CHECK(row_begin, "row_begin"); NULL_CHECK(row_end, "row_end");
    if (dev->split_world > 1 && dev->split_layout != MIRHI_SPLIT_BANDS)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: the device's tile split is interleaved (no single band of rows): use mirhi_device_split_rows");
    uint32_t first, step, count;
    split_rows(MIRHI_SPLIT_BANDS, dev->split_rank, dev->split_world, (height + TILE - 1) / TILE, &first, &step, &count);
    uint32_t p0 = first * TILE, p1 = (first + count) * TILE;
    if (p0 > height) p0 = height;
    if (p1 > height) p1 = height;
    *row_begin = p0; *row_end = p1;
    return MIRHI_OK;
}
static const char* usage_name(mirhi_buffer_usage u) {
    static const char* names[] = {"vertex", "index", "uniform", "storage", "staging", "indirect"};
    return (u >= 0 && u <= 5) ? names[u] : "unknown";
}
static bool usage_host_visible(mirhi_buffer_usage u) {   // BufferUsage::memory_location buffer.rs:86-100
    return u == MIRHI_BUFFER_VERTEX || u == MIRHI_BUFFER_INDEX || u == MIRHI_BUFFER_UNIFORM || u == MIRHI_BUFFER_STAGING;
}

extern "C" mirhi_result mirhi_buffer_create(mirhi_device* dev, mirhi_buffer_usage usage, uint64_t size, mirhi_buffer** out) {
    NULL_CHECK(dev, "device"); NULL_CHECK(out, "out");
    *out = nullptr;
    if ((int)usage < 0 || (int)usage > 5) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: unknown buffer usage %d", (int)usage);
    if (size == 0) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: Buffer size must be greater than 0");   // buffer.rs:150-154
    HIP_TRY(hipSetDevice(dev->ordinal));
    void* p = nullptr;
    // Uniform buffers are what a frame loop rewrites every frame (Buffer::write_data on mapped memory, buffer.rs:247-279): like the
    // parameter block they live in fine-grained device memory that the host writes with plain stores (0.2 us per KB, no HIP call).
    // MIRHI_PARAM_UPLOAD=copy (or a refused allocation) keeps them in plain device memory behind copies.
    const char* upload = getenv("MIRHI_PARAM_UPLOAD");
    const bool want_copy = upload && strcmp(upload, "copy") == 0;
    bool direct = false;
    hipError_t e = hipErrorNotSupported;
    if (usage == MIRHI_BUFFER_UNIFORM && size <= (1u << 20) && !want_copy) {
        e = hipExtMallocWithFlags(&p, (size + 255) & ~(uint64_t)255, hipDeviceMallocFinegrained);
        direct = e == hipSuccess;
        if (!direct) (void)hipGetLastError();
    }
    if (!direct) e = hipMalloc(&p, (size + 255) & ~(uint64_t)255);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: hipMalloc(%llu) for %s buffer: %s", (unsigned long long)size, usage_name(usage), hipGetErrorString(e)); }
    mirhi_buffer* b = new (std::nothrow) mirhi_buffer{dev, usage, size, (uint8_t*)p, true};
    if (!b) { (void)hipFree(p); return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: host allocation failed"); }
    b->host_direct = direct;
    dev->children++;
    *out = b;
    return MIRHI_OK;
}
// Waits for the submissions that may still READ `buf` -- and for no others: the command buffers whose recording names memory inside the
// buffer and that are pending.  (In the reference the memory is mapped and write_data is a memcpy: not writing what a frame in flight
// reads is the caller's business, which is why it keeps one uniform buffer per frame in flight.  A frame loop that has waited for its
// slot's fence finds nothing pending here; a caller that has not is made to wait for exactly the frames concerned.)
static mirhi_result settle_pending(mirhi_cmd* cmd, bool in_submit = false);
static mirhi_result settle_readers(mirhi_device* dev, const uint8_t* lo, const uint8_t* hi);
extern "C" mirhi_result mirhi_buffer_write(mirhi_buffer* buf, uint64_t offset, const void* data, uint64_t len) {
    NULL_CHECK(buf, "buffer");
    if (len == 0) return MIRHI_OK;                                                        // buffer.rs:248-250
    NULL_CHECK(data, "data");
    if (offset + len > buf->size || offset + len < offset)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: Write exceeds buffer size: offset %llu + data %llu > buffer %llu",
                    (unsigned long long)offset, (unsigned long long)len, (unsigned long long)buf->size);   // buffer.rs:252-260
    if (!usage_host_visible(buf->usage))
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: Buffer memory is not mapped");              // buffer.rs:266-268
    // host-coherent write semantics: ordered after the submitted work that reads the buffer, visible to later submits
    { mirhi_result r0 = settle_readers(buf->dev, buf->ptr, buf->ptr + buf->size); if (r0 != MIRHI_OK) return r0; }
    if (buf->host_direct) {
        memcpy(buf->ptr + offset, data, len);           // write-combined stores over the BAR ...
        __builtin_ia32_sfence();                          // ... out of the write-combining buffers before a later submit rings a doorbell
        return MIRHI_OK;
    }
    HIP_TRY(hipSetDevice(buf->dev->ordinal));
    buf->dev->foreign_writes++;
    HIP_TRY(hipMemcpyAsync(buf->ptr + offset, data, len, hipMemcpyHostToDevice, buf->dev->stream));
    HIP_TRY(hipStreamSynchronize(buf->dev->stream));
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_buffer_upload(mirhi_buffer* buf, const void* data, uint64_t len) { return mirhi_buffer_write(buf, 0, data, len); }
extern "C" mirhi_result mirhi_buffer_upload_via_staging(mirhi_buffer* buf, const void* data, uint64_t len) {
    NULL_CHECK(buf, "buffer");
    if (len == 0) return MIRHI_OK;
    NULL_CHECK(data, "data");
    if (len > buf->size)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: Upload exceeds buffer size: data %llu > buffer %llu", (unsigned long long)len, (unsigned long long)buf->size);
    { mirhi_result r0 = sync_all_lanes(buf->dev); if (r0 != MIRHI_OK) return r0; }
    buf->dev->foreign_writes++;
    HIP_TRY(hipMemcpyAsync(buf->ptr, data, len, hipMemcpyHostToDevice, buf->dev->stream));
    HIP_TRY(hipStreamSynchronize(buf->dev->stream));
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_buffer_create_with_data(mirhi_device* dev, mirhi_buffer_usage usage, const void* data, uint64_t len, mirhi_buffer** out) {
    mirhi_result r = mirhi_buffer_create(dev, usage, len, out);                           // buffer.rs:227-231
    if (r != MIRHI_OK) return r;
    r = mirhi_buffer_write(*out, 0, data, len);
    if (r != MIRHI_OK) { std::string keep = g_last_error; mirhi_buffer_destroy(*out); *out = nullptr; g_last_error = keep; }
    return r;
}
extern "C" mirhi_result mirhi_buffer_wrap_device_memory(mirhi_device* dev, mirhi_buffer_usage usage, void* device_ptr, uint64_t size, mirhi_buffer** out) {
    NULL_CHECK(dev, "device"); NULL_CHECK(out, "out");
    *out = nullptr;
    if (size == 0 || !device_ptr) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: Buffer size must be greater than 0");
    mirhi_buffer* b = new (std::nothrow) mirhi_buffer{dev, usage, size, (uint8_t*)device_ptr, false};
    if (!b) return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: host allocation failed");
    dev->foreign_writes++;
    dev->children++;
    *out = b;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_buffer_read(mirhi_buffer* buf, uint64_t offset, void* dst, uint64_t len) {
    NULL_CHECK(buf, "buffer");
    if (len == 0) return MIRHI_OK;
    NULL_CHECK(dst, "dst");
    if (offset + len > buf->size || offset + len < offset)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: Read exceeds buffer size: offset %llu + len %llu > buffer %llu",
                    (unsigned long long)offset, (unsigned long long)len, (unsigned long long)buf->size);
    { mirhi_result r0 = sync_all_lanes(buf->dev); if (r0 != MIRHI_OK) return r0; }
    HIP_TRY(hipMemcpyAsync(dst, buf->ptr + offset, len, hipMemcpyDeviceToHost, buf->dev->stream));
    HIP_TRY(hipStreamSynchronize(buf->dev->stream));
    return MIRHI_OK;
}
extern "C" uint64_t mirhi_buffer_size(const mirhi_buffer* buf) { return buf ? buf->size : 0; }
extern "C" int32_t mirhi_buffer_usage_of(const mirhi_buffer* buf) { return buf ? (int32_t)buf->usage : -1; }
extern "C" void* mirhi_buffer_device_ptr(const mirhi_buffer* buf) { return buf ? buf->ptr : nullptr; }
extern "C" mirhi_result mirhi_buffer_destroy(mirhi_buffer* buf) {
    NULL_CHECK(buf, "buffer");
    (void)hipSetDevice(buf->dev->ordinal);
    if (buf->owned) { (void)sync_all_lanes(buf->dev); (void)hipFree(buf->ptr); }
    buf->dev->children--;
    delete buf;
    return MIRHI_OK;
}

// ------------------------------------------------------------------------------------------------
// images
// ------------------------------------------------------------------------------------------------
static mirhi_result image_common(mirhi_device* dev, uint32_t w, uint32_t h, mirhi_format f, void* ext, mirhi_image** out) {
    NULL_CHECK(dev, "device"); NULL_CHECK(out, "out");
    *out = nullptr;
    if (w == 0 || h == 0)   // depth_buffer.rs:118-127
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: Image dimensions must be greater than 0 (got %ux%u)", w, h);
    if (w > 16384 || h > 16384) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: Image dimensions exceed 16384 (got %ux%u)", w, h);
    const uint32_t bpp = format_bpp(f);
    if (bpp == 0) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: unsupported image format %d", (int)f);
    void* p = ext;
    if (!ext) {
        HIP_TRY(hipSetDevice(dev->ordinal));
        hipError_t e = hipMalloc(&p, (size_t)w * h * bpp);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: hipMalloc for %ux%u image: %s", w, h, hipGetErrorString(e)); }
    }
    mirhi_image* img = new (std::nothrow) mirhi_image{dev, w, h, f, (uint8_t*)p, ext == nullptr};
    if (!img) { if (!ext) (void)hipFree(p); return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: host allocation failed"); }
    { std::lock_guard<std::mutex> lock(dev->mu); dev->images.push_back(img); }
    dev->children++;
    *out = img;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_image_create(mirhi_device* dev, uint32_t w, uint32_t h, mirhi_format f, mirhi_image** out) { return image_common(dev, w, h, f, nullptr, out); }
extern "C" mirhi_result mirhi_image_wrap_device_memory(mirhi_device* dev, uint32_t w, uint32_t h, mirhi_format f, void* ptr, mirhi_image** out) {
    if (!ptr) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: device_ptr is null");
    if (dev) dev->foreign_writes++;
    return image_common(dev, w, h, f, ptr, out);
}
extern "C" uint32_t mirhi_image_width(const mirhi_image* img) { return img ? img->width : 0; }
extern "C" uint32_t mirhi_image_height(const mirhi_image* img) { return img ? img->height : 0; }
extern "C" int32_t mirhi_image_format(const mirhi_image* img) { return img ? (int32_t)img->format : 0; }
extern "C" uint64_t mirhi_image_size_bytes(const mirhi_image* img) { return img ? (uint64_t)img->width * img->height * format_bpp(img->format) : 0; }
extern "C" void* mirhi_image_device_ptr(const mirhi_image* img) { return img ? img->ptr : nullptr; }
extern "C" mirhi_result mirhi_image_upload(mirhi_image* img, const void* src, uint64_t len) {
    NULL_CHECK(img, "image"); NULL_CHECK(src, "src");
    if (len != mirhi_image_size_bytes(img))
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: image upload size %llu != image size %llu", (unsigned long long)len, (unsigned long long)mirhi_image_size_bytes(img));
    { mirhi_result r0 = sync_all_lanes(img->dev); if (r0 != MIRHI_OK) return r0; }
    img->dev->foreign_writes++;
    HIP_TRY(hipMemcpyAsync(img->ptr, src, len, hipMemcpyHostToDevice, img->dev->stream));
    HIP_TRY(hipStreamSynchronize(img->dev->stream));
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_image_read(mirhi_image* img, void* dst, uint64_t len) {
    NULL_CHECK(img, "image"); NULL_CHECK(dst, "dst");
    if (len != mirhi_image_size_bytes(img))
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: image read size %llu != image size %llu", (unsigned long long)len, (unsigned long long)mirhi_image_size_bytes(img));
    { mirhi_result r0 = sync_all_lanes(img->dev); if (r0 != MIRHI_OK) return r0; }
    HIP_TRY(hipMemcpyAsync(dst, img->ptr, len, hipMemcpyDeviceToHost, img->dev->stream));
    HIP_TRY(hipStreamSynchronize(img->dev->stream));
    return MIRHI_OK;
}
// one thread per texel of the destination level: 2x2 box filter on the stored bytes, round half up, edge clamp
__global__ void mip_kernel(const uint32_t* __restrict__ src, uint32_t sw, uint32_t sh, uint32_t* __restrict__ dst, uint32_t dw, uint32_t dh) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    const uint32_t x0 = 2u * x < sw ? 2u * x : sw - 1u, x1 = 2u * x + 1u < sw ? 2u * x + 1u : sw - 1u;
    const uint32_t y0 = 2u * y < sh ? 2u * y : sh - 1u, y1 = 2u * y + 1u < sh ? 2u * y + 1u : sh - 1u;
    const uint32_t a = src[y0 * sw + x0], b = src[y0 * sw + x1], c = src[y1 * sw + x0], e = src[y1 * sw + x1];
    uint32_t out = 0;
    for (uint32_t sft = 0; sft < 32u; sft += 8u)
        out |= ((((a >> sft) & 0xFFu) + ((b >> sft) & 0xFFu) + ((c >> sft) & 0xFFu) + ((e >> sft) & 0xFFu) + 2u) >> 2) << sft;
    dst[y * dw + x] = out;
}
extern "C" uint32_t mirhi_image_mip_levels(const mirhi_image* img) { return img ? img->levels : 0; }
extern "C" mirhi_result mirhi_image_generate_mips(mirhi_image* img) {
    NULL_CHECK(img, "image");
    if (img->format != MIRHI_FORMAT_R8G8B8A8_UNORM && img->format != MIRHI_FORMAT_R8G8B8A8_SRGB)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: mip chains are built for R8G8B8A8 textures only");
    if (!img->owned) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: cannot grow a wrapped image into a mip chain");
    HIP_TRY(hipSetDevice(img->dev->ordinal));
    { mirhi_result r0 = sync_all_lanes(img->dev); if (r0 != MIRHI_OK) return r0; }
    uint32_t levels = 1;
    size_t texels = (size_t)img->width * img->height;
    for (uint32_t w = img->width, h = img->height; w > 1 || h > 1; levels++) { w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1; texels += (size_t)w * h; }
    uint8_t* chain = img->ptr;
    if (img->levels != levels) {              // first call: move level 0 into a buffer that holds the whole chain
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, texels * 4);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: hipMalloc for a %zu-texel mip chain: %s", texels, hipGetErrorString(e)); }
        chain = (uint8_t*)p;
        img->dev->foreign_writes++;
        HIP_TRY(hipMemcpyAsync(chain, img->ptr, (size_t)img->width * img->height * 4, hipMemcpyDeviceToDevice, img->dev->stream));
    }
    uint32_t sw = img->width, sh = img->height;
    uint32_t* src = (uint32_t*)chain;
    for (uint32_t l = 1; l < levels; l++) {
        const uint32_t dw = sw > 1 ? sw >> 1 : 1, dh = sh > 1 ? sh >> 1 : 1;
        uint32_t* dst = src + (size_t)sw * sh;
        hipLaunchKernelGGL(mip_kernel, dim3((dw + 63) / 64, dh), dim3(64), 0, img->dev->stream, src, sw, sh, dst, dw, dh);
        HIP_TRY(hipGetLastError());
        src = dst; sw = dw; sh = dh;
    }
    HIP_TRY(hipStreamSynchronize(img->dev->stream));
    if (chain != img->ptr) { (void)hipFree(img->ptr); img->ptr = chain; }
    img->levels = levels;
    return MIRHI_OK;
}
extern "C" uint32_t mirhi_image_max_anisotropy(const mirhi_image* img) { return img ? img->max_anisotropy : 0; }
extern "C" mirhi_result mirhi_image_set_max_anisotropy(mirhi_image* img, uint32_t max_anisotropy) {
    NULL_CHECK(img, "image");
    if (img->format != MIRHI_FORMAT_R8G8B8A8_UNORM && img->format != MIRHI_FORMAT_R8G8B8A8_SRGB)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: anisotropic filtering applies to sampled R8G8B8A8 textures only");
    if (max_anisotropy < 1u || max_anisotropy > 16u)
        return fail(MIRHI_ERR_DEVICE, "Vulkan error: maxAnisotropy %u outside [1, 16] (maxSamplerAnisotropy)", max_anisotropy);
    img->max_anisotropy = max_anisotropy;        // read when a draw is recorded, like the mip chain
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_image_destroy(mirhi_image* img) {
    NULL_CHECK(img, "image");
    (void)hipSetDevice(img->dev->ordinal);
    if (img->owned) { (void)sync_all_lanes(img->dev); (void)hipFree(img->ptr); }
    { std::lock_guard<std::mutex> lock(img->dev->mu); auto& v = img->dev->images; v.erase(std::remove(v.begin(), v.end(), img), v.end()); }
    img->dev->children--;
    delete img;
    return MIRHI_OK;
}

// ------------------------------------------------------------------------------------------------
// pipeline (pipeline.rs:645-698 defaults, :918-952 validation)
// ------------------------------------------------------------------------------------------------
extern "C" void mirhi_pipeline_desc_default(mirhi_pipeline_desc* d) {
    if (!d) return;
    memset(d, 0, sizeof *d);
    d->vertex_program = MIRHI_PROGRAM_NONE;
    d->fragment_program = MIRHI_PROGRAM_NONE;
    d->topology = MIRHI_TOPOLOGY_TRIANGLE_LIST;
    d->polygon_mode = MIRHI_POLYGON_FILL;
    d->cull_mode = MIRHI_CULL_BACK;
    d->front_face = MIRHI_FRONT_FACE_COUNTER_CLOCKWISE;
    d->rasterization_samples = 1;
    d->depth_test_enable = 1;
    d->depth_write_enable = 1;
    d->depth_compare_op = MIRHI_COMPARE_LESS;
    d->depth_attachment_format = MIRHI_FORMAT_UNDEFINED;
    d->src_color_blend_factor = MIRHI_BLEND_ONE; d->dst_color_blend_factor = MIRHI_BLEND_ZERO; d->color_blend_op = MIRHI_BLEND_OP_ADD;   // pipeline.rs:499-512
    d->src_alpha_blend_factor = MIRHI_BLEND_ONE; d->dst_alpha_blend_factor = MIRHI_BLEND_ZERO; d->alpha_blend_op = MIRHI_BLEND_OP_ADD;
    d->color_write_mask = 0xFu;
}

extern "C" mirhi_result mirhi_pipeline_create(mirhi_device* dev, const mirhi_pipeline_desc* d, mirhi_pipeline** out) {
    NULL_CHECK(dev, "device"); NULL_CHECK(d, "desc"); NULL_CHECK(out, "out");
    *out = nullptr;
    // --- the reference's own validation, same order and text (pipeline.rs:920-952) ---
    if (d->vertex_program == MIRHI_PROGRAM_NONE) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: Vertex shader is required");
    if (d->fragment_program == MIRHI_PROGRAM_NONE) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: Fragment shader is required");
    if (d->color_attachment_count == 0) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: At least one color attachment format is required");
    const bool has_depth = d->depth_attachment_format != MIRHI_FORMAT_UNDEFINED;
    if ((d->depth_test_enable || d->depth_write_enable) && !has_depth)
        return fail(MIRHI_ERR_PIPELINE, "Pipeline error: Depth test or write is enabled but no depth attachment format is specified");
    if (d->blend_attachment_count != 0 && d->blend_attachment_count != d->color_attachment_count)
        return fail(MIRHI_ERR_PIPELINE, "Pipeline error: Blend attachment count (%u) must match color attachment count (%u)", d->blend_attachment_count, d->color_attachment_count);
    // --- program selection replaces SPIR-V module creation (shader.rs:244-330) ---
    if (d->vertex_program < 0 || d->vertex_program > MIRHI_PROGRAM_MODEL_PBR || d->fragment_program < 0 || d->fragment_program > MIRHI_PROGRAM_MODEL_PBR)
        return fail(MIRHI_ERR_SHADER, "Shader error: unknown program id (vertex %d, fragment %d)", d->vertex_program, d->fragment_program);
    const bool vs_model = d->vertex_program != MIRHI_PROGRAM_TRIANGLE, fs_model = d->fragment_program != MIRHI_PROGRAM_TRIANGLE;
    if (vs_model != fs_model)
        return fail(MIRHI_ERR_SHADER, "Shader error: vertex program %d does not produce the inputs of fragment program %d", d->vertex_program, d->fragment_program);
    // --- what this rasterizer does not implement fails loudly instead of rendering something else ---
    if (d->color_attachment_count != 1) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: unsupported: %u color attachments (exactly 1 supported)", d->color_attachment_count);
    const int32_t cf = d->color_attachment_formats[0];
    if (cf != MIRHI_FORMAT_B8G8R8A8_SRGB && cf != MIRHI_FORMAT_R32G32B32A32_SFLOAT)
        return fail(MIRHI_ERR_PIPELINE, "Pipeline error: unsupported color attachment format %d", cf);
    if (has_depth && d->depth_attachment_format != MIRHI_FORMAT_D32_SFLOAT)
        return fail(MIRHI_ERR_PIPELINE, "Pipeline error: unsupported depth attachment format %d (D32_SFLOAT only)", d->depth_attachment_format);
    if (d->topology != MIRHI_TOPOLOGY_TRIANGLE_LIST) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: unsupported topology %d (TriangleList only)", d->topology);
    if (d->polygon_mode != MIRHI_POLYGON_FILL) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: unsupported polygon mode %d (Fill only)", d->polygon_mode);
    if (d->cull_mode < 0 || d->cull_mode > 3 || d->front_face < 0 || d->front_face > 1) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: invalid cull mode / front face");
    if (d->depth_compare_op < 0 || d->depth_compare_op > 7) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: invalid depth compare op %d", d->depth_compare_op);
    if (d->rasterization_samples != 1) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: unsupported sample count %u", d->rasterization_samples);
    if (d->blend_enable) {
        const int32_t f[4] = {d->src_color_blend_factor, d->dst_color_blend_factor, d->src_alpha_blend_factor, d->dst_alpha_blend_factor};
        for (int32_t v : f) {
            if (v < 0 || v > MIRHI_BLEND_SRC_ALPHA_SATURATE) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: invalid blend factor %d", v);
            if (v >= MIRHI_BLEND_CONSTANT_COLOR && v <= MIRHI_BLEND_ONE_MINUS_CONSTANT_ALPHA)
                return fail(MIRHI_ERR_PIPELINE, "Pipeline error: unsupported: constant-colour blend factors (no blend constants on this path)");
        }
        if (d->color_blend_op < 0 || d->color_blend_op > MIRHI_BLEND_OP_MAX || d->alpha_blend_op < 0 || d->alpha_blend_op > MIRHI_BLEND_OP_MAX)
            return fail(MIRHI_ERR_PIPELINE, "Pipeline error: invalid blend op");
    }
    if (d->depth_clamp_enable || d->depth_bias_enable) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: unsupported: depth clamp / depth bias");
    const uint32_t want_stride = vs_model ? 48u : 24u;      // vertex.rs:35-41 / :130-136
    if (d->vertex_stride < want_stride || (d->vertex_stride & 3u))
        return fail(MIRHI_ERR_PIPELINE, "Pipeline error: vertex stride %u too small for program %d (needs >= %u, multiple of 4)", d->vertex_stride, d->vertex_program, want_stride);
    const uint32_t want_attrs = vs_model ? 4u : 2u;
    static const uint32_t model_offsets[4] = {0, 12, 24, 32}, tri_offsets[2] = {0, 12};
    if (d->attribute_count != want_attrs) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: program %d expects %u vertex attributes, got %u", d->vertex_program, want_attrs, d->attribute_count);
    for (uint32_t i = 0; i < want_attrs; i++)
        if (d->attribute_offsets[i] != (vs_model ? model_offsets[i] : tri_offsets[i]))
            return fail(MIRHI_ERR_PIPELINE, "Pipeline error: attribute %u offset %u does not match the reference vertex layout", i, d->attribute_offsets[i]);
    mirhi_pipeline* p = new (std::nothrow) mirhi_pipeline{dev, *d};
    if (!p) return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: host allocation failed");
    dev->children++;
    *out = p;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_pipeline_destroy(mirhi_pipeline* p) {
    NULL_CHECK(p, "pipeline");
    p->dev->children--;
    delete p;
    return MIRHI_OK;
}

// ------------------------------------------------------------------------------------------------
// command recording (command.rs)
// ------------------------------------------------------------------------------------------------
extern "C" void mirhi_rendering_info_default(mirhi_rendering_info* i) {
    if (!i) return;
    memset(i, 0, sizeof *i);
    i->color_load_op = MIRHI_LOAD_OP_CLEAR; i->color_store_op = MIRHI_STORE_OP_STORE;       // rendering.rs:102-115
    i->clear_color[3] = 1.0f;
    i->depth_load_op = MIRHI_LOAD_OP_CLEAR; i->depth_store_op = MIRHI_STORE_OP_DONT_CARE;   // rendering.rs:356-370
    i->clear_depth = 1.0f;
}

static void free_workspace(mirhi_cmd* c) {
    Workspace& w = c->ws;
    if (w.pblock) (void)hipFree(w.pblock);
    if (w.pstage) (void)hipHostFree(w.pstage);
    if (w.bin_pool) (void)hipFree(w.bin_pool);
    if (w.bin_table) (void)hipFree(w.bin_table);
    if (w.counters) (void)hipFree(w.counters);
    if (w.big_recs) (void)hipFree(w.big_recs);
    if (w.carry_depth) (void)hipFree(w.carry_depth);
    if (w.ordered) (void)hipFree(w.ordered);
    if (w.vs_out) (void)hipFree(w.vs_out);
    if (w.flat_color) (void)hipFree(w.flat_color);
    if (w.prim_draw) (void)hipFree(w.prim_draw);
    if (w.stats_prim) (void)hipFree(w.stats_prim);
    if (w.stats_params) (void)hipFree(w.stats_params);
    if (w.status_host) (void)hipHostFree(w.status_host);
    w = Workspace();
}

// The device-side status of a finished submission lives in the command buffer's workspace.  Before that workspace goes away
// (destroy) or is re-armed (end() of a new recording) every fence that still lists the command buffer -- Vulkan allows
// destroying a command buffer before its fence is waited on, and the Rust wrapper's Drop order does exactly that -- takes the
// status over; so does the device's wait_idle bookkeeping.  The caller has synchronised the command buffer's lane.
static mirhi_result status_of(mirhi_device* dev, mirhi_cmd* c);
static void hand_over_status(mirhi_cmd* cmd) {
    mirhi_device* dev = cmd->dev;
    std::lock_guard<std::mutex> lock(dev->mu);
    auto& u = dev->unchecked;
    const bool was_unchecked = std::find(u.begin(), u.end(), cmd) != u.end();
    bool listed = false;
    for (mirhi_fence* f : dev->fences) listed |= std::find(f->cmds.begin(), f->cmds.end(), cmd) != f->cmds.end();
    if (!was_unchecked && !listed) return;
    const std::string keep = g_last_error;
    const mirhi_result rc = status_of(dev, cmd);
    const std::string msg = g_last_error;
    g_last_error = keep;
    for (mirhi_fence* f : dev->fences) {
        bool had = false;
        for (size_t i = f->cmds.size(); i-- > 0;)
            if (f->cmds[i] == cmd) { f->cmds.erase(f->cmds.begin() + (ptrdiff_t)i); f->seqs.erase(f->seqs.begin() + (ptrdiff_t)i); had = true; }
        if (had && rc != MIRHI_OK) { f->deferred = rc; f->deferred_msg = msg; }
    }
    if (was_unchecked) {
        u.erase(std::remove(u.begin(), u.end(), cmd), u.end());
        if (rc != MIRHI_OK && !listed) { dev->deferred = rc; dev->deferred_msg = msg; }
    }
}

extern "C" mirhi_result mirhi_cmd_create(mirhi_device* dev, mirhi_cmd** out) {
    NULL_CHECK(dev, "device"); NULL_CHECK(out, "out");
    mirhi_cmd* c = new (std::nothrow) mirhi_cmd();
    if (!c) return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: host allocation failed");
    c->dev = dev;
    c->lane = dev->next_lane++ % (uint32_t)dev->lanes.size();
    { std::lock_guard<std::mutex> lock(dev->mu); dev->cmds.push_back(c); }
    dev->children++;
    *out = c;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_set_queue_lane(mirhi_cmd* cmd, uint32_t lane) {
    NULL_CHECK(cmd, "command buffer");
    if (lane >= cmd->dev->lanes.size()) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: queue lane %u of %zu", lane, cmd->dev->lanes.size());
    drain_submits(cmd->dev);
    if (!drain_native(cmd->dev, cmd->last_native)) return device_lost(cmd->dev);
    if (cmd->last_stream) HIP_TRY(hipStreamSynchronize(cmd->last_stream));
    HIP_TRY(hipStreamSynchronize(cmd->dev->lanes[cmd->lane < cmd->dev->lanes.size() ? cmd->lane : 0]));
    cmd_finished(cmd);
    cmd->lane = lane;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_destroy(mirhi_cmd* cmd) {
    NULL_CHECK(cmd, "command buffer");
    (void)sync_all_lanes(cmd->dev);
    cmd_finished(cmd);
    hand_over_status(cmd);          // fences (and wait_idle) that still list this command buffer keep its device status
    { std::lock_guard<std::mutex> lock(cmd->dev->mu); auto& v = cmd->dev->cmds; v.erase(std::remove(v.begin(), v.end(), cmd), v.end()); }
    free_workspace(cmd);
    cmd->dev->children--;
    delete cmd;
    return MIRHI_OK;
}
static void reset_recording(mirhi_cmd* c) {
    // (the launch plan and the recording it was built from stay: end() compares the new recording with it -- see mirhi_cmd::planned)
    c->passes.clear();
    c->in_rendering = false;
    c->pipeline = nullptr; c->vb = nullptr; c->ib = nullptr;
    for (auto& u : c->uniforms) u = {nullptr, 0, 0};
    for (auto& tx : c->textures) tx = nullptr;
    c->has_viewport = c->has_scissor = false;
}
static mirhi_result begin_common(mirhi_cmd* cmd, bool one_time) {
    NULL_CHECK(cmd, "command buffer");
    if (cmd->state == CMD_RECORDING) return fail(MIRHI_ERR_DEVICE, "Vulkan error: command buffer is already recording");
    while (cmd->queued.load(std::memory_order_acquire) > 0) cpu_relax();
    reset_recording(cmd);          // implicit reset, as with RESET_COMMAND_BUFFER pools (command.rs:89-106)
    cmd->state = CMD_RECORDING;
    cmd->one_time = one_time;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_begin(mirhi_cmd* cmd) { return begin_common(cmd, true); }
extern "C" mirhi_result mirhi_cmd_begin_reusable(mirhi_cmd* cmd) { return begin_common(cmd, false); }
extern "C" mirhi_result mirhi_cmd_reset(mirhi_cmd* cmd) {
    NULL_CHECK(cmd, "command buffer");
    while (cmd->queued.load(std::memory_order_acquire) > 0) cpu_relax();
    reset_recording(cmd);
    cmd->state = CMD_INITIAL;
    return MIRHI_OK;
}
#define REQUIRE_RECORDING(cmd)                                                                         \
    do {                                                                                               \
        NULL_CHECK(cmd, "command buffer");                                                             \
        if ((cmd)->state != CMD_RECORDING) return fail(MIRHI_ERR_DEVICE, "Vulkan error: command buffer is not in the recording state"); \
    } while (0)

extern "C" mirhi_result mirhi_cmd_begin_rendering(mirhi_cmd* cmd, const mirhi_rendering_info* info) {
    REQUIRE_RECORDING(cmd); NULL_CHECK(info, "rendering_info");
    if (cmd->in_rendering) return fail(MIRHI_ERR_DEVICE, "Vulkan error: begin_rendering inside an active rendering scope");
    NULL_CHECK(info->color_image, "color_image");
    const mirhi_image* ci = info->color_image;
    if (ci->format != MIRHI_FORMAT_B8G8R8A8_SRGB && ci->format != MIRHI_FORMAT_R32G32B32A32_SFLOAT)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: color attachment has non-colour format %d", (int)ci->format);
    if (info->depth_image) {
        if (info->depth_image->format != MIRHI_FORMAT_D32_SFLOAT) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: depth attachment is not D32_SFLOAT");
        if (info->depth_image->width != ci->width || info->depth_image->height != ci->height)
            return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: depth attachment extent does not match the colour attachment");
    }
    if (info->prim_id_image) {
        if (info->prim_id_image->format != MIRHI_FORMAT_R32_UINT || info->prim_id_image->width != ci->width || info->prim_id_image->height != ci->height)
            return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: prim_id_image must be R32_UINT with the colour attachment's extent");
    }
    RecordedPass p;
    p.info = *info;
    p.color_t = {ci->ptr, ci->width, ci->height, (uint32_t)ci->format};
    if (info->depth_image) p.depth_t = {info->depth_image->ptr, info->depth_image->width, info->depth_image->height, (uint32_t)info->depth_image->format};
    if (info->prim_id_image) p.prim_t = {info->prim_id_image->ptr, info->prim_id_image->width, info->prim_id_image->height, (uint32_t)info->prim_id_image->format};
    if (info->render_area[2] <= 0 || info->render_area[3] <= 0) { p.area[0] = 0; p.area[1] = 0; p.area[2] = (int32_t)ci->width; p.area[3] = (int32_t)ci->height; }
    else memcpy(p.area, info->render_area, sizeof p.area);
    if (p.area[0] != 0 || p.area[1] != 0 || p.area[2] != (int32_t)ci->width || p.area[3] != (int32_t)ci->height)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: unsupported: render area must cover the whole colour attachment");
    cmd->passes.push_back(std::move(p));
    cmd->in_rendering = true;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_end_rendering(mirhi_cmd* cmd) {
    REQUIRE_RECORDING(cmd);
    if (!cmd->in_rendering) return fail(MIRHI_ERR_DEVICE, "Vulkan error: end_rendering without begin_rendering");
    cmd->in_rendering = false;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_bind_pipeline(mirhi_cmd* cmd, mirhi_pipeline* pipeline) {
    REQUIRE_RECORDING(cmd); NULL_CHECK(pipeline, "pipeline");
    cmd->pipeline = pipeline;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_bind_vertex_buffers(mirhi_cmd* cmd, uint32_t first_binding, uint32_t count, mirhi_buffer* const* buffers, const uint64_t* offsets) {
    REQUIRE_RECORDING(cmd);
    if (count == 0) return MIRHI_OK;
    NULL_CHECK(buffers, "buffers"); NULL_CHECK(offsets, "offsets");
    if (first_binding != 0 || count != 1) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: only vertex binding 0 exists (vertex.rs:35-41)");
    NULL_CHECK(buffers[0], "buffers[0]");
    if (offsets[0] >= buffers[0]->size || (offsets[0] & 3)) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: vertex buffer offset %llu out of range or unaligned", (unsigned long long)offsets[0]);
    cmd->vb = buffers[0]; cmd->vb_offset = offsets[0];
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_bind_index_buffer(mirhi_cmd* cmd, mirhi_buffer* buffer, uint64_t offset, mirhi_index_type type) {
    REQUIRE_RECORDING(cmd); NULL_CHECK(buffer, "buffer");
    if (type != MIRHI_INDEX_UINT16 && type != MIRHI_INDEX_UINT32) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: unknown index type %d", (int)type);
    const uint64_t isz = type == MIRHI_INDEX_UINT16 ? 2 : 4;
    if (offset >= buffer->size || (offset % isz)) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: index buffer offset %llu out of range or unaligned", (unsigned long long)offset);
    cmd->ib = buffer; cmd->ib_offset = offset; cmd->ib_type = type;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_bind_uniform(mirhi_cmd* cmd, mirhi_uniform_slot slot, mirhi_buffer* buffer, uint64_t offset, uint64_t range) {
    REQUIRE_RECORDING(cmd); NULL_CHECK(buffer, "buffer");
    if ((int)slot < 0 || (int)slot >= MIRHI_SLOT_COUNT) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: unknown uniform slot %d", (int)slot);
    if (range == 0 || range == UINT64_MAX) range = buffer->size > offset ? buffer->size - offset : 0;   // VK_WHOLE_SIZE
    if (offset + range > buffer->size || range == 0 || (offset & 15))
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: uniform range offset %llu + range %llu exceeds buffer %llu or offset not 16-byte aligned",
                    (unsigned long long)offset, (unsigned long long)range, (unsigned long long)buffer->size);
    cmd->uniforms[slot] = {buffer, offset, range};
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_bind_texture(mirhi_cmd* cmd, mirhi_texture_slot slot, mirhi_image* image) {
    REQUIRE_RECORDING(cmd);
    if ((int)slot < 0 || (int)slot >= MIRHI_TEXTURE_COUNT) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: unknown texture slot %d", (int)slot);
    if (image && image->format != MIRHI_FORMAT_R8G8B8A8_UNORM && image->format != MIRHI_FORMAT_R8G8B8A8_SRGB)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: sampled images must be R8G8B8A8_UNORM or R8G8B8A8_SRGB");
    cmd->textures[slot] = image;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_set_viewport(mirhi_cmd* cmd, const mirhi_viewport* vp) {
    REQUIRE_RECORDING(cmd); NULL_CHECK(vp, "viewport");
    if (!(vp->width > 0.0f) || !(vp->height != 0.0f) || !(vp->min_depth >= 0.0f && vp->min_depth <= 1.0f) || !(vp->max_depth >= 0.0f && vp->max_depth <= 1.0f))
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: viewport width must be > 0, height != 0, depth range within [0,1]");
    if (std::fabs(vp->x) + std::fabs(vp->width) > 8192.0f || std::fabs(vp->y) + std::fabs(vp->height) > 8192.0f)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: viewport exceeds the +-8192 px range supported by the guard band");
    cmd->viewport = *vp; cmd->has_viewport = true;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_set_scissor(mirhi_cmd* cmd, const mirhi_rect2d* sc) {
    REQUIRE_RECORDING(cmd); NULL_CHECK(sc, "scissor");
    if (sc->x < 0 || sc->y < 0) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: scissor offset must be non-negative");
    cmd->scissor = *sc; cmd->has_scissor = true;
    return MIRHI_OK;
}

static mirhi_result record_draw(mirhi_cmd* cmd, bool indexed, uint32_t count, uint32_t instance_count, uint32_t first, int32_t vertex_offset) {
    REQUIRE_RECORDING(cmd);
    if (!cmd->in_rendering) return fail(MIRHI_ERR_DEVICE, "Vulkan error: draw outside a rendering scope");
    if (!cmd->pipeline) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: no pipeline bound");
    if (!cmd->vb) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: no vertex buffer bound to binding 0");
    if (indexed && !cmd->ib) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: no index buffer bound");
    if (!cmd->has_viewport || !cmd->has_scissor) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: viewport and scissor are dynamic state and must be set before drawing (pipeline.rs:697)");
    if (instance_count > 1) {
        // The path has no instance-rate input (binding 0 is per-vertex, vertex.rs:35-41,130-136; no program reads SV_InstanceID): instance i
        // rasterizes the same triangles again, behind instance i - 1 in primitive order -- recorded as that many draws of one instance.
        if (instance_count > 4096u) return fail(MIRHI_ERR_PIPELINE, "Pipeline error: unsupported: instance_count %u (at most 4096 per draw)", instance_count);
        for (uint32_t inst = 0; inst < instance_count; inst++) {
            const mirhi_result ri = record_draw(cmd, indexed, count, 1u, first, vertex_offset);
            if (ri != MIRHI_OK) return ri;
        }
        return MIRHI_OK;
    }
    const mirhi_pipeline_desc& pd = cmd->pipeline->desc;
    const mirhi_image* ci = cmd->passes.back().info.color_image;
    if (pd.color_attachment_formats[0] != (int32_t)ci->format)
        return fail(MIRHI_ERR_PIPELINE, "Pipeline error: pipeline colour format %d does not match the attachment format %d", pd.color_attachment_formats[0], (int)ci->format);
    // depth state must be uniform within a rendering scope (DESIGN.md "Depth key")
    const uint32_t dtest = pd.depth_test_enable ? 1u : 0u;
    const uint32_t dcmp = dtest ? (uint32_t)pd.depth_compare_op : (uint32_t)MIRHI_COMPARE_ALWAYS;
    const uint32_t dwrite = dtest && pd.depth_write_enable ? 1u : 0u;       // Vulkan: no depth write without the depth test
    const bool never = pd.depth_test_enable && pd.depth_compare_op == MIRHI_COMPARE_NEVER;
    const uint32_t tri_count = count / 3u;
    if (instance_count == 0 || tri_count == 0) return MIRHI_OK;
    if (never) { cmd->passes.back().total_tris += tri_count; return MIRHI_OK; }     // draws nothing, but its primitives keep their ids
    uint32_t blend[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pd.blend_enable) {
        blend[0] = 1; blend[1] = (uint32_t)pd.src_color_blend_factor; blend[2] = (uint32_t)pd.dst_color_blend_factor; blend[3] = (uint32_t)pd.color_blend_op;
        blend[4] = (uint32_t)pd.src_alpha_blend_factor; blend[5] = (uint32_t)pd.dst_alpha_blend_factor; blend[6] = (uint32_t)pd.alpha_blend_op;
        blend[7] = pd.color_write_mask & 0xFu;
    }
    const uint32_t discard = pd.fragment_discard_enable ? 1u : 0u;
    if (!cmd->passes.back().key_set) {
        RecordedPass& p0 = cmd->passes.back();
        p0.key_set = true; p0.depth_test = dtest; p0.depth_compare = dcmp; p0.depth_write = dwrite; p0.frag_discard = discard;
        memcpy(p0.blend, blend, sizeof blend);
    } else if (cmd->passes.back().depth_test != dtest || cmd->passes.back().depth_compare != dcmp || cmd->passes.back().depth_write != dwrite ||
               cmd->passes.back().frag_discard != discard || memcmp(cmd->passes.back().blend, blend, sizeof blend) != 0) {
        // One raster launch resolves one depth state (DESIGN.md "Depth key"): the scope continues in a new segment that
        // loads what the previous one stored -- fragments keep their submission order across the cut.
        RecordedPass next;
        {
            RecordedPass& prev = cmd->passes.back();
            prev.carry_out = true;
            next.info = prev.info;
            next.color_t = prev.color_t; next.depth_t = prev.depth_t; next.prim_t = prev.prim_t;
            memcpy(next.area, prev.area, sizeof next.area);
            next.info.color_load_op = MIRHI_LOAD_OP_LOAD;
            next.carry_in = true;
            next.first_tri = next.total_tris = prev.total_tris;
        }
        next.key_set = true; next.depth_test = dtest; next.depth_compare = dcmp; next.depth_write = dwrite; next.frag_discard = discard;
        memcpy(next.blend, blend, sizeof blend);
        cmd->passes.push_back(std::move(next));
    }
    RecordedPass& pass = cmd->passes.back();

    DrawDesc d;
    memset(&d, 0, sizeof d);
    d.stride = pd.vertex_stride;
    d.vb = cmd->vb->ptr + cmd->vb_offset;
    const uint64_t vb_bytes = cmd->vb->size - cmd->vb_offset;
    const uint32_t vsize = pd.vertex_program == MIRHI_PROGRAM_TRIANGLE ? 24u : 48u;
    if (!indexed) {
        const uint64_t last = (uint64_t)first + 3ull * tri_count;   // one past the last vertex read
        if ((last - 1) * d.stride + vsize > vb_bytes)
            return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: draw reads vertices [%u, %llu) beyond the bound vertex buffer (%llu bytes)", first, (unsigned long long)last, (unsigned long long)vb_bytes);
        d.index_type = 0; d.first = first; d.vertex_offset = 0; d.ib = nullptr;
    } else {
        const uint64_t isz = cmd->ib_type == MIRHI_INDEX_UINT16 ? 2 : 4;
        const uint64_t ib_bytes = cmd->ib->size - cmd->ib_offset;
        if (((uint64_t)first + 3ull * tri_count) * isz > ib_bytes)
            return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: draw_indexed reads indices [%u, %llu) beyond the bound index buffer", first, (unsigned long long)first + 3ull * tri_count);
        d.index_type = (uint32_t)isz; d.first = first; d.vertex_offset = vertex_offset; d.ib = cmd->ib->ptr + cmd->ib_offset;
    }
    d.tri_count = tri_count;
    d.prim_base = pass.total_tris;
    if ((uint64_t)pass.total_tris + tri_count > (uint64_t)MAX_PRIM_ID) return fail(MIRHI_ERR_DEVICE, "Vulkan error: too many primitives in one rendering scope");
    d.program = (uint32_t)pd.fragment_program;
    d.cull_mode = (uint32_t)pd.cull_mode; d.front_face = (uint32_t)pd.front_face;
    // uniforms required by the program (model.hlsl:5-19, model_full.hlsl:27-50)
    auto uptr = [&](int slot, uint64_t need, const uint8_t** out, const char* name) -> mirhi_result {
        auto& u = cmd->uniforms[slot];
        if (!u.buf) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: program %d needs %s bound", pd.fragment_program, name);
        if (u.range < need) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: %s range %llu smaller than %llu bytes", name, (unsigned long long)u.range, (unsigned long long)need);
        *out = u.buf->ptr + u.offset;
        return MIRHI_OK;
    };
    if (d.program != MIRHI_PROGRAM_TRIANGLE) {
        const uint8_t* p = nullptr;
        mirhi_result r;
        if ((r = uptr(MIRHI_SLOT_CAMERA, 208, &p, "CameraData (b0)")) != MIRHI_OK) return r;
        d.camera = (const float*)p;
        if ((r = uptr(MIRHI_SLOT_OBJECT, 128, &p, "ObjectData (b1)")) != MIRHI_OK) return r;
        d.object = (const float*)p;
        if (d.program == MIRHI_PROGRAM_MODEL_FULL || d.program == MIRHI_PROGRAM_MODEL_PBR) {
            if ((r = uptr(MIRHI_SLOT_LIGHTS, 48, &d.lights, "LightUBO (b2)")) != MIRHI_OK) return r;
            if ((r = uptr(MIRHI_SLOT_MATERIAL, d.program == MIRHI_PROGRAM_MODEL_PBR ? 80 : 32, &d.material, "MaterialData (b3)")) != MIRHI_OK) return r;
            if (cmd->uniforms[MIRHI_SLOT_POINT_LIGHTS].buf) d.point_lights = cmd->uniforms[MIRHI_SLOT_POINT_LIGHTS].buf->ptr + cmd->uniforms[MIRHI_SLOT_POINT_LIGHTS].offset;
            if (cmd->uniforms[MIRHI_SLOT_SPOT_LIGHTS].buf) d.spot_lights = cmd->uniforms[MIRHI_SLOT_SPOT_LIGHTS].buf->ptr + cmd->uniforms[MIRHI_SLOT_SPOT_LIGHTS].offset;
            for (int t = 0; t < (d.program == MIRHI_PROGRAM_MODEL_PBR ? 5 : 2); t++)
                if (cmd->textures[t]) {
                    const mirhi_image* ti = cmd->textures[t];
                    d.tex[t] = ti->ptr; d.tex_w[t] = ti->width; d.tex_h[t] = ti->height; d.tex_levels[t] = ti->levels;
                    if (ti->format == MIRHI_FORMAT_R8G8B8A8_SRGB) d.tex_srgb |= 1u << t;
                    if (ti->levels > 1) d.tex_any_mips = 1;
                    if (ti->levels > 1) d.tex_aniso |= (ti->max_anisotropy - 1u) << (4 * t);
                }
        }
    }
    // viewport (Vulkan: xf = (w/2) xd + (x + w/2)), guard-band factors, scissor
    const mirhi_viewport& vp = cmd->viewport;
    d.hw = 0.5f * vp.width; d.hh = 0.5f * vp.height;
    d.cx = vp.x + d.hw; d.cy = vp.y + d.hh;
    d.dscale = vp.max_depth - vp.min_depth; d.dmin = vp.min_depth;
    d.gx = (GUARD_PX - std::fabs(d.cx)) / d.hw;
    d.gy = (GUARD_PX - std::fabs(d.cy)) / d.hh;
    int64_t sx0 = cmd->scissor.x, sy0 = cmd->scissor.y;
    int64_t sx1 = sx0 + (int64_t)cmd->scissor.width - 1, sy1 = sy0 + (int64_t)cmd->scissor.height - 1;
    if (sx1 > (int64_t)ci->width - 1) sx1 = (int64_t)ci->width - 1;
    if (sy1 > (int64_t)ci->height - 1) sy1 = (int64_t)ci->height - 1;
    d.sx0 = (int32_t)sx0; d.sy0 = (int32_t)sy0; d.sx1 = (int32_t)sx1; d.sy1 = (int32_t)sy1;
    d.scissor_partial = (sx0 > 0 || sy0 > 0 || sx1 < (int64_t)ci->width - 1 || sy1 < (int64_t)ci->height - 1) ? 1u : 0u;
    if (sx0 > sx1 || sy0 > sy1) return MIRHI_OK;    // empty scissor: nothing can be covered
    pass.draws.push_back(d);
    pass.draw_vb_bytes.push_back(vb_bytes);
    pass.total_tris += tri_count;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_draw(mirhi_cmd* cmd, uint32_t vertex_count, uint32_t instance_count, uint32_t first_vertex, uint32_t first_instance) {
    (void)first_instance;
    return record_draw(cmd, false, vertex_count, instance_count, first_vertex, 0);
}
extern "C" mirhi_result mirhi_cmd_draw_indexed(mirhi_cmd* cmd, uint32_t index_count, uint32_t instance_count, uint32_t first_index, int32_t vertex_offset, uint32_t first_instance) {
    (void)first_instance;
    return record_draw(cmd, true, index_count, instance_count, first_index, vertex_offset);
}

// draw_indirect / draw_indexed_indirect (command.rs:630-661): the arguments are fetched from the device buffer now, at record time
static mirhi_result record_indirect(mirhi_cmd* cmd, bool indexed, mirhi_buffer* buffer, uint64_t offset, uint32_t draw_count, uint32_t stride) {
    REQUIRE_RECORDING(cmd); NULL_CHECK(buffer, "buffer");
    if (buffer->dev != cmd->dev) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: indirect buffer belongs to another device");
    const uint32_t words = indexed ? 5u : 4u;
    if (draw_count == 0) return MIRHI_OK;
    if ((offset & 3u) || (draw_count > 1 && (stride < words * 4u || (stride & 3u))))
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: indirect offset %llu / stride %u must be multiples of 4 (stride >= %u)", (unsigned long long)offset, stride, words * 4u);
    const uint64_t step = draw_count > 1 ? stride : words * 4u;
    const uint64_t span = (uint64_t)(draw_count - 1) * step + words * 4u;
    if (offset + span > buffer->size || offset + span < offset)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: indirect draws read [%llu, %llu) beyond the buffer (%llu bytes)", (unsigned long long)offset,
                    (unsigned long long)(offset + span), (unsigned long long)buffer->size);
    std::vector<uint8_t> host(span);
    { mirhi_result r0 = sync_all_lanes(cmd->dev); if (r0 != MIRHI_OK) return r0; }
    HIP_TRY(hipMemcpyAsync(host.data(), buffer->ptr + offset, span, hipMemcpyDeviceToHost, cmd->dev->stream));
    HIP_TRY(hipStreamSynchronize(cmd->dev->stream));
    for (uint32_t i = 0; i < draw_count; i++) {
        uint32_t a[5] = {0, 0, 0, 0, 0};
        memcpy(a, host.data() + (size_t)i * step, words * 4u);
        const mirhi_result r = indexed ? record_draw(cmd, true, a[0], a[1], a[2], (int32_t)a[3]) : record_draw(cmd, false, a[0], a[1], a[2], 0);
        if (r != MIRHI_OK) return r;
    }
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_cmd_draw_indirect(mirhi_cmd* cmd, mirhi_buffer* buffer, uint64_t offset, uint32_t draw_count, uint32_t stride) {
    return record_indirect(cmd, false, buffer, offset, draw_count, stride);
}
extern "C" mirhi_result mirhi_cmd_draw_indexed_indirect(mirhi_cmd* cmd, mirhi_buffer* buffer, uint64_t offset, uint32_t draw_count, uint32_t stride) {
    return record_indirect(cmd, true, buffer, offset, draw_count, stride);
}
extern "C" mirhi_result mirhi_cmd_push_constants(mirhi_cmd* cmd, uint32_t stage_flags, uint32_t offset, const void* data, uint32_t len) {
    REQUIRE_RECORDING(cmd);
    (void)stage_flags;
    if (len == 0) return MIRHI_OK;
    NULL_CHECK(data, "data");
    if ((offset & 3u) || (len & 3u) || (uint64_t)offset + len > sizeof cmd->push_constants)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: push constants offset %u + size %u must be multiples of 4 within %zu bytes", offset, len, sizeof cmd->push_constants);
    memcpy(cmd->push_constants + offset, data, len);
    return MIRHI_OK;
}

// ---- end(): size the workspace and build the launch plan ------------------------------------------
template <typename T>
static mirhi_result grow(T** ptr, size_t* have, size_t want_bytes) {
    if (*have >= want_bytes && *ptr) return MIRHI_OK;
    if (*ptr) { (void)hipFree(*ptr); *ptr = nullptr; *have = 0; }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want_bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: hipMalloc(%zu) for rasterizer workspace: %s", want_bytes, hipGetErrorString(e)); }
    *ptr = (T*)p; *have = want_bytes;
    return MIRHI_OK;
}

// A segment is resolved fragment by fragment in primitive order (ordered_kernel) when its colour is blended, when its
// depth state makes the stored depth depend on the order of all fragments (NotEqual with depth write), or when its fragment
// program may discard single fragments (mirhi_pipeline_desc::fragment_discard_enable: visibility then needs the program's result).
static void depth_key_setup(PassParams& P, const RecordedPass& pass);
// fragment_discard_enable without blending under a depth state the depth key resolves by minimum (not a predicate state): visibility stays
// order-independent -- a kept fragment competes by its key -- so the segment keeps bins and the raster kernel, and alpha is tested per
// covered pixel in front of the key minimum (PassParams::alpha_scope, raster_small_masked).  MIRHI_MASKED_ORDERED=1 (A/B runs): the
// ordered resolve instead.
static bool pass_is_masked_plain(const RecordedPass& pass) {
    if (!pass.key_set || !pass.frag_discard || pass.blend[0] != 0) return false;
    if (pass.depth_test && pass.depth_write && pass.depth_compare == MIRHI_COMPARE_NOT_EQUAL) return false;
    if (getenv("MIRHI_MASKED_ORDERED") && atoi(getenv("MIRHI_MASKED_ORDERED")) != 0) return false;
    PassParams key{};
    depth_key_setup(key, pass);
    return key.pred == 0u;
}
static bool pass_is_ordered(const RecordedPass& pass) {
    return pass.key_set && (pass.blend[0] != 0 || (pass.frag_discard != 0 && !pass_is_masked_plain(pass)) ||
                            (pass.depth_test && pass.depth_write && pass.depth_compare == MIRHI_COMPARE_NOT_EQUAL));
}

static void depth_key_setup(PassParams& P, const RecordedPass& pass) {
    const uint32_t cbits = [&] { float f = pass.info.clear_depth; f = f > 0.0f ? (f < 1.0f ? f : 1.0f) : 0.0f; uint32_t u; memcpy(&u, &f, 4); return u; }();
    P.clear_depth_bits = cbits;
    P.pred = 0;
    const uint32_t op = pass.key_set ? pass.depth_compare : (uint32_t)MIRHI_COMPARE_ALWAYS;
    const bool test = pass.key_set && pass.depth_test;
    const bool write = test && pass.depth_write;
    if (!test || (op == MIRHI_COMPARE_ALWAYS && !write)) {      // every fragment passes, nothing is written: later primitive wins
        P.zflip = 0; P.zmask = 0; P.idflip = 1; P.strict = 0; P.init_zk = 0; P.init_idk = NO_PRIM; return;
    }
    const bool ordered = write && (op == MIRHI_COMPARE_LESS || op == MIRHI_COMPARE_LESS_OR_EQUAL || op == MIRHI_COMPARE_GREATER ||
                                   op == MIRHI_COMPARE_GREATER_OR_EQUAL);
    if (!ordered) {
        // Predicate mode.  Without depth write (or with Equal, which can only rewrite the same value) the stored depth
        // never changes inside the scope: every fragment is tested against the depth the scope started with and the latest
        // passing primitive owns the pixel.  Always with write: everything passes, the latest primitive's depth is stored.
        static const uint32_t bits[8] = {0u, 1u, 2u, 3u, 4u, 5u, 6u, 7u};     // Never, Less, Equal, LessOrEqual, Greater, NotEqual, GreaterOrEqual, Always
        P.pred = bits[op & 7u] | (op == MIRHI_COMPARE_ALWAYS ? 8u : 0u);
        if (P.pred == 0) P.pred = 16u;                                           // (Never is dropped at record time; keep the mode bit set)
        P.zflip = 0; P.zmask = 0xFFFFFFFFu; P.idflip = 1; P.strict = 0;
        P.init_zk = cbits; P.init_idk = NO_PRIM;
        return;
    }
    const bool greater = (op == MIRHI_COMPARE_GREATER || op == MIRHI_COMPARE_GREATER_OR_EQUAL);
    P.strict = (op == MIRHI_COMPARE_LESS || op == MIRHI_COMPARE_GREATER) ? 1u : 0u;
    P.zflip = greater ? 0xFFFFFFFFu : 0u;
    P.zmask = 0xFFFFFFFFu;
    P.idflip = P.strict ? 0u : 1u;            // strict: earlier primitive keeps ties; or-equal: later primitive wins
    const uint32_t t = cbits ^ P.zflip;
    if (!P.strict) { P.init_zk = t; P.init_idk = NO_PRIM; }
    else if (t == 0u) { P.init_zk = 0u; P.init_idk = 0u; }   // nothing can pass
    else { P.init_zk = t - 1u; P.init_idk = NO_PRIM; }
}

// How a scope is rastered (decided once per recorded scope, used for sizing the workspace and for the launch):
//  tp_max_area  triangle-parallel resolve of small records pays when tiles hold many triangles (meshes); sparse scopes keep
//               the leaner pixel-parallel-only kernel.  Scopes of TRIANGLE-program draws switch at 16 triangles per tile on
//               average (their variant gives up one wave of occupancy for the LDS key array); mesh-program scopes lose nothing
//               and a mesh covers a fraction of the frame (the dancer asset: 8 per tile on average, 124 per tile it touches),
//               so they switch at 4.  Box limit 64 pixels: measured against 96 / 128 on the dancer (74 / 81 / 89 us), C3
//               (38.6 / 36.8 / 37.1), C4 (112.9 / 112.4 / 112.4) and C5 (202 / 206 / 211).
//  teams        two teams per tile + per-XCD bins when a mesh scope is dense enough for the triangle-parallel variant yet
//               averages under 16 triangles per tile: then its triangles sit in a small part of the frame (the dancer: 919 in
//               the fullest tile), the chip is far from full, the raster kernel lasts as long as the fullest tile's serial
//               chain -- which two teams cut (dancer raster 62.6 -> 44.0 us; four teams: 47.4) -- and the geometry kernel as
//               long as the queue of atomics on the hottest bin counter, which per-XCD counters cut (see reserve_bin_slots).
//  MIRHI_TP_MAX_AREA (0 = off), MIRHI_TP_DENSITY, MIRHI_RASTER_TEAMS (1 / 2) override for A/B measurements.
struct RasterMode { uint32_t tp_max_area, teams; bool tri_prog; bool wide_eligible; bool xcd_bins; uint32_t wide; };
// wide: what the command buffer's busy-tile feedback asks for (Workspace::wide: 0 / 8 / 16 waves per tile)
static RasterMode raster_mode(const RecordedPass& pass, size_t tiles, bool spread = false, uint32_t wide = 0) {
    RasterMode m{0u, 1u, false, false, false, 0u};
    for (const DrawDesc& dd : pass.draws) m.tri_prog |= dd.program == MIRHI_PROGRAM_TRIANGLE;
    PassParams key{};
    depth_key_setup(key, pass);
    const size_t avg = tiles ? (pass.total_tris - pass.first_tri) / tiles : 0;
    const size_t density = getenv("MIRHI_TP_DENSITY") ? (size_t)atoi(getenv("MIRHI_TP_DENSITY")) : (m.tri_prog ? 16 : 4);
    const bool dense = tiles && avg >= density;
    m.tp_max_area = getenv("MIRHI_TP_MAX_AREA") ? (uint32_t)atoi(getenv("MIRHI_TP_MAX_AREA")) : (dense ? 64u : 0u);
    if (key.pred) m.tp_max_area = 0;      // predicate scopes resolve pixel-parallel only (the LDS key array holds ordered keys)
    else if (pass_is_masked_plain(pass) && m.tp_max_area == 0u) m.tp_max_area = 1u;    // alpha-masked scope: its records need the triangle-parallel path (LDS key array)
    const bool mesh_only = !m.tri_prog && !pass.draws.empty();
    m.teams = getenv("MIRHI_RASTER_TEAMS") ? (uint32_t)atoi(getenv("MIRHI_RASTER_TEAMS")) : (avg < 16 ? 2u : 1u);
    if (!(m.tp_max_area && mesh_only && !pass_is_ordered(pass)) || m.teams != 2u) m.teams = 1u;
    if (spread && !getenv("MIRHI_RASTER_TEAMS")) m.teams = 1u;      // measured on an earlier submission of this command buffer (Workspace::spread)
    // per-XCD bins go with the concentrated-mesh mode (the geometry kernel's counter contention), whichever raster variant then reads them
    m.xcd_bins = m.teams == 2u && !(getenv("MIRHI_XCD_BINS") && atoi(getenv("MIRHI_XCD_BINS")) == 0);
    // The wide variants (eight / sixteen waves per tile, raster_body WPT) take over from both the plain and the two-team variant once the
    // busy-tile count says the mesh sits in few tiles.  MIRHI_RASTER_WIDE = 0 / 8 / 16 forces it (tests, A/B runs); a forced
    // MIRHI_RASTER_TEAMS = 2 keeps the two teams.
    m.wide_eligible = m.tp_max_area && mesh_only && !pass_is_ordered(pass) && !pass_is_masked_plain(pass) && !key.pred &&
                      !(getenv("MIRHI_RASTER_TEAMS") && atoi(getenv("MIRHI_RASTER_TEAMS")) == 2 && !getenv("MIRHI_RASTER_WIDE"));
    if (m.wide_eligible) {
        const uint32_t forced = getenv("MIRHI_RASTER_WIDE") ? (uint32_t)atoi(getenv("MIRHI_RASTER_WIDE")) : 0xFFFFFFFFu;
        m.wide = forced == 0xFFFFFFFFu ? wide : (forced == 0u ? 0u : (forced == 8u ? 8u : 16u));
        // (teams stays what the scope gets when a submit decides against the wide variant: see mirhi_queue_submit, "frames in flight")
    }
    return m;
}

// Sizes the workspace of a recorded command buffer and builds its launch plan.  Runs at end() -- unless the recording is the one the
// current plan was built from -- and again in front of a submit when an earlier submission exhausted the bin pool
// (Workspace::grow_pool) or showed a spread-out mesh (Workspace::replan).
static mirhi_result build_plan(mirhi_cmd* cmd, bool in_submit = false);
static mirhi_result pblock_commit(Workspace& w, hipStream_t stream);

static bool same_target(const RecordedPass::Target& a, const RecordedPass::Target& b) {
    return a.ptr == b.ptr && a.width == b.width && a.height == b.height && a.format == b.format;
}
// Two recordings describe the same frame shape: same attachments, load / store ops, clear values, depth / blend state and, byte for
// byte, the same draw descriptors (resolved device pointers, counts, viewport, scissor, state).  Buffer CONTENTS are not part of it:
// the plan holds pointers, and a frame loop that rewrites its uniform buffer between frames records the same shape every time.
// any_color_address: the colour targets may sit at other addresses (same extent and format) -- the recording of a frame loop that cycles N + 1 swapchain images
// through N command buffers (swapchain.rs:228-236, renderer.rs:377-390): everything the plan holds but PassParams::color is the same then
static bool same_recording(const std::vector<RecordedPass>& a, const std::vector<RecordedPass>& b, bool any_color_address = false) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++) {
        const RecordedPass& x = a[i]; const RecordedPass& y = b[i];
        if (any_color_address ? (x.color_t.width != y.color_t.width || x.color_t.height != y.color_t.height || x.color_t.format != y.color_t.format || !x.color_t.ptr != !y.color_t.ptr)
                              : !same_target(x.color_t, y.color_t)) return false;
        if (!same_target(x.depth_t, y.depth_t) || !same_target(x.prim_t, y.prim_t)) return false;
        if (x.info.color_load_op != y.info.color_load_op || x.info.color_store_op != y.info.color_store_op || x.info.depth_load_op != y.info.depth_load_op ||
            x.info.depth_store_op != y.info.depth_store_op || memcmp(x.info.clear_color, y.info.clear_color, sizeof x.info.clear_color) != 0 ||
            memcmp(&x.info.clear_depth, &y.info.clear_depth, sizeof(float)) != 0) return false;
        if (x.total_tris != y.total_tris || x.first_tri != y.first_tri || x.key_set != y.key_set || x.depth_test != y.depth_test || x.depth_compare != y.depth_compare ||
            x.depth_write != y.depth_write || x.frag_discard != y.frag_discard || memcmp(x.blend, y.blend, sizeof x.blend) != 0 ||
            x.carry_in != y.carry_in || x.carry_out != y.carry_out || memcmp(x.area, y.area, sizeof x.area) != 0) return false;
        if (x.draws.size() != y.draws.size() || x.draw_vb_bytes != y.draw_vb_bytes) return false;
        if (!x.draws.empty() && memcmp(x.draws.data(), y.draws.data(), x.draws.size() * sizeof(DrawDesc)) != 0) return false;
    }
    return true;
}

// The workspace of a command buffer that may still be executing must not be touched: Vulkan forbids re-recording a pending command
// buffer, this build waits for it.  A frame loop that waits on its in-flight fence first (renderer.rs:371-374) never waits here.
static mirhi_result settle_pending(mirhi_cmd* cmd, bool in_submit) {
    // (submit thread on: not before its queued submissions have been issued -- unless this IS the submit thread, re-planning the job it holds)
    while (!in_submit && cmd->queued.load(std::memory_order_acquire) > 0) cpu_relax();
    if (!cmd->pending) return MIRHI_OK;
    mirhi_device* dev = cmd->dev;
    if (cmd->last_native && !drain_native(dev, cmd->last_native)) return device_lost(dev);
    bool live = false;
    for (hipStream_t st : dev->lanes) live |= st == cmd->last_stream;
    if (live) HIP_TRY(hipStreamSynchronize(cmd->last_stream));
    cmd_finished(cmd);
    return MIRHI_OK;
}

static bool recording_reads(const std::vector<RecordedPass>& passes, const uint8_t* lo, const uint8_t* hi) {
    auto in = [&](const void* q) { return q && (const uint8_t*)q >= lo && (const uint8_t*)q < hi; };
    for (const RecordedPass& pass : passes)
        for (const DrawDesc& d : pass.draws) {
            if (in(d.vb) || in(d.ib) || in(d.camera) || in(d.object) || in(d.lights) || in(d.material) || in(d.point_lights) || in(d.spot_lights)) return true;
        }
    return false;
}
static mirhi_result settle_readers(mirhi_device* dev, const uint8_t* lo, const uint8_t* hi) {
    std::vector<mirhi_cmd*> readers;
    {
        std::lock_guard<std::mutex> lk(dev->mu);
        for (mirhi_cmd* c : dev->cmds)
            // (`planned` is the recording the pending submissions were planned from and does not change while one is pending; `passes` may be
            // under re-recording by another host thread at this very moment)
            if ((c->pending || c->queued.load(std::memory_order_acquire) > 0) && recording_reads(c->planned, lo, hi)) readers.push_back(c);
    }
    for (mirhi_cmd* c : readers) { const mirhi_result r = settle_pending(c, false); if (r != MIRHI_OK) return r; }
    return MIRHI_OK;
}

#ifdef MIRHI_HOST_PROF
#include <x86intrin.h>
static unsigned long long g_hp[8]; static unsigned long long g_hp_n;
#define HP(k) do { const unsigned long long t_ = __rdtsc(); g_hp[k] += t_ - hp_t; hp_t = t_; } while (0)
#define HP_BEGIN unsigned long long hp_t = __rdtsc(); g_hp_n++
extern "C" void mirhi_debug_host_prof(void) { for (int k = 0; k < 8; k++) fprintf(stderr, "host prof [%d] %.1f cycles per call (%llu calls)\n", k, g_hp_n ? (double)g_hp[k] / (double)g_hp_n : 0.0, g_hp_n); }
#else
#define HP(k) do {} while (0)
#define HP_BEGIN do {} while (0)
#endif
extern "C" mirhi_result mirhi_cmd_end(mirhi_cmd* cmd) {
    HP_BEGIN;
    REQUIRE_RECORDING(cmd);
    if (cmd->in_rendering) return fail(MIRHI_ERR_DEVICE, "Vulkan error: end() inside an active rendering scope");
    mirhi_device* dev = cmd->dev;
    if (cmd->plan_valid && !cmd->ws.grow_pool && !cmd->ws.replan && !cmd->ws.dirty && cmd->plan_split_rank == dev->split_rank && cmd->plan_split_world == dev->split_world && cmd->plan_split_layout == dev->split_layout &&
        same_recording(cmd->passes, cmd->planned)) {
        HP(0);
        // The frame recorded last time, recorded again: plan, workspace and parameter block are what they have to be.  No HIP call,
        // no synchronisation, nothing uploaded; only the status words of the previous submission are handed over and re-armed.
        mirhi_result r = settle_pending(cmd);
        if (r != MIRHI_OK) return r;
        HP(1);
        if (cmd->ws.status_host && (cmd->ws.status_host[0] | cmd->ws.status_host[1] | cmd->ws.status_host[2] | cmd->ws.status_host[3])) {
            HP(2);
            hand_over_status(cmd);
            HP(3);
            cmd->ws.status_host[0] = 0; cmd->ws.status_host[1] = 0; cmd->ws.status_host[2] = 0; cmd->ws.status_host[3] = 0;
            HP(4);
        }
        if (!(cmd->ws.dirty || cmd->ws.replan || cmd->ws.grow_pool)) {      // (the status just handed over may have asked for clears or another plan: rebuild below)
            cmd->state = CMD_EXECUTABLE;
            return MIRHI_OK;
        }
    }
    if (cmd->plan_valid && !cmd->ws.grow_pool && !cmd->ws.replan && !cmd->ws.dirty && cmd->plan_split_rank == dev->split_rank && cmd->plan_split_world == dev->split_world && cmd->plan_split_layout == dev->split_layout &&
        cmd->plan.size() == cmd->passes.size() && cmd->ws.pimage.size() >= cmd->passes.size() * 2 * sizeof(PassParams) && same_recording(cmd->passes, cmd->planned, true)) {
        // The same frame into another swapchain image: the plan stands, PassParams::color of every scope is rewritten (host copy, both parities in the parameter block).
        mirhi_result r = settle_pending(cmd);                  // (the block is read by a submission that is still pending: wait for it -- a fenced loop has)
        if (r != MIRHI_OK) return r;
        if (cmd->ws.status_host && (cmd->ws.status_host[0] | cmd->ws.status_host[1] | cmd->ws.status_host[2] | cmd->ws.status_host[3])) {
            hand_over_status(cmd);
            cmd->ws.status_host[0] = 0; cmd->ws.status_host[1] = 0; cmd->ws.status_host[2] = 0; cmd->ws.status_host[3] = 0;
        }
        if (cmd->ws.replan || cmd->ws.grow_pool || cmd->ws.dirty) goto rebuild;      // (the status just handed over asked for another plan)
        {
            HIP_TRY(hipSetDevice(dev->ordinal));
            hipStream_t stream = dev->lanes[cmd->lane < dev->lanes.size() ? cmd->lane : 0];
            for (size_t pi = 0; pi < cmd->passes.size(); pi++) {
                uint8_t* target = const_cast<uint8_t*>(cmd->passes[pi].color_t.ptr);
                cmd->plan[pi].color = target;
                for (size_t parity = 0; parity < 2; parity++) reinterpret_cast<PassParams*>(cmd->ws.pimage.data() + (2 * pi + parity) * sizeof(PassParams))->color = target;
            }
            if ((r = pblock_commit(cmd->ws, stream)) != MIRHI_OK) { cmd->plan_valid = false; return r; }
            if (!cmd->ws.pblock_direct) { cmd->last_stream = stream; cmd->last_native = nullptr; cmd->pending = true; }     // (the copy is in the lane's stream)
            { std::lock_guard<std::mutex> lk(dev->mu); cmd->planned = cmd->passes; }
            cmd->state = CMD_EXECUTABLE;
            return MIRHI_OK;
        }
    }
rebuild:
    mirhi_result r = build_plan(cmd);
    if (r != MIRHI_OK) { cmd->plan_valid = false; return r; }
    cmd->state = CMD_EXECUTABLE;
    return MIRHI_OK;
}

// Host-written parameter block (Workspace::pblock).  Fine-grained device memory is mapped into the host's address space; if the
// platform refuses either the allocation or the mapping, MIRHI_PARAM_UPLOAD=copy (or a failed allocation) selects plain device
// memory filled by stream-ordered copies from a pinned staging buffer.
static mirhi_result pblock_reserve(Workspace& w, size_t bytes) {
    if (w.pblock && w.pblock_bytes >= bytes) return MIRHI_OK;
    const size_t want = (bytes + 4095) & ~(size_t)4095;
    if (w.pblock) { (void)hipFree(w.pblock); w.pblock = nullptr; w.pblock_bytes = 0; }
    if (w.pstage) { (void)hipHostFree(w.pstage); w.pstage = nullptr; w.pstage_bytes = 0; }
    w.pshadow.clear();
    void* p = nullptr;
    const bool want_copy = getenv("MIRHI_PARAM_UPLOAD") && strcmp(getenv("MIRHI_PARAM_UPLOAD"), "copy") == 0;
    hipError_t e = want_copy ? hipErrorNotSupported : hipExtMallocWithFlags(&p, want, hipDeviceMallocFinegrained);
    w.pblock_direct = e == hipSuccess;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipMalloc(&p, want);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: hipMalloc(%zu) for the parameter block: %s", want, hipGetErrorString(e)); }
        void* h = nullptr;
        e = hipHostMalloc(&h, want, hipHostMallocDefault);
        if (e != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: hipHostMalloc(%zu) for parameter staging: %s", want, hipGetErrorString(e)); }
        w.pstage = (uint8_t*)h; w.pstage_bytes = want;
    }
    w.pblock = (uint8_t*)p; w.pblock_bytes = want;
    return MIRHI_OK;
}
// Brings the block to `pimage`: nothing if it holds those bytes already.
static mirhi_result pblock_commit(Workspace& w, hipStream_t stream) {
    const size_t n = w.pimage.size();
    if (w.pshadow.size() == n && (n == 0 || memcmp(w.pshadow.data(), w.pimage.data(), n) == 0)) return MIRHI_OK;
    if (n) {
        if (w.pblock_direct) {
            // write-combined stores over the BAR, only the 64-byte pieces that differ from what the block holds (a frame loop changes a target address or a count) ...
            if (w.pshadow.size() == n) {
                for (size_t o = 0; o < n; o += 64) {
                    const size_t len = n - o < 64 ? n - o : 64;
                    if (memcmp(w.pshadow.data() + o, w.pimage.data() + o, len) != 0) memcpy(w.pblock + o, w.pimage.data() + o, len);
                }
            } else memcpy(w.pblock, w.pimage.data(), n);
            __builtin_ia32_sfence();                         // ... out of the write-combining buffers before any launch rings a doorbell
        } else {
            memcpy(w.pstage, w.pimage.data(), n);
            HIP_TRY(hipMemcpyAsync(w.pblock, w.pstage, n, hipMemcpyHostToDevice, stream));
            w.foreign = true;
        }
    }
    w.pshadow = w.pimage;
    return MIRHI_OK;
}

static mirhi_result build_plan(mirhi_cmd* cmd, bool in_submit) {
    mirhi_device* dev = cmd->dev;
    HIP_TRY(hipSetDevice(dev->ordinal));
    if (cmd->lane >= dev->lanes.size()) cmd->lane = 0;
    hipStream_t stream = dev->lanes[cmd->lane];
    cmd->plan_valid = false;
    { mirhi_result rs = settle_pending(cmd, in_submit); if (rs != MIRHI_OK) return rs; }
    {
        uint64_t tris_now = 0;
        for (auto& pass : cmd->passes) tris_now += pass.total_tris - pass.first_tri;
        if (cmd->ws.spread && (tris_now > 2 * cmd->ws.spread_tris || 2 * tris_now < cmd->ws.spread_tris)) cmd->ws.spread = false;
    }
    size_t total_draws = 0, max_tiles = 0, max_pages = 0, max_big = 0;
    struct Geo { uint32_t tiles_x, tiles_y, r0, r1, rstep, bin_cap, sub_cap, big_cap, fixed_pages, fixed_per_tile; bool xcd_bins; };      // r0 / r1 / rstep: PassParams::tile_row_begin / _end / _step
    std::vector<Geo> geo;
    for (auto& pass : cmd->passes) {
        const RecordedPass::Target& ci = pass.color_t;
        Geo g;
        g.tiles_x = (ci.width + TILE - 1) / TILE; g.tiles_y = (ci.height + TILE - 1) / TILE;
        { uint32_t count; split_rows(dev->split_layout, dev->split_rank, dev->split_world, g.tiles_y, &g.r0, &g.rstep, &count); g.r1 = g.r0 + count; }
        const size_t tiles = (size_t)g.tiles_x * (g.r1 - g.r0);
        const RasterMode mode = raster_mode(pass, tiles, cmd->ws.spread, cmd->ws.wide);
        g.xcd_bins = mode.xcd_bins;
        // A tile's bin holds up to BIN_TABLE_ROW pages (4096 records; eight lists of 512 with per-XCD bins) before it spills into
        // the big list, which EVERY tile walks -- the limit costs nothing until it is used: pages come out of one pool, sized by
        // the scope's triangle count, not by tiles x capacity (round 1: 100 MB at 1080p, 400-510 MB at 4K per command buffer).
        // MIRHI_BIN_CAP (records per list, A/B runs and the spill tests) lowers it.
        uint32_t cap = (uint32_t)BIN_TABLE_ROW * BIN_PAGE_RECS;
        if (getenv("MIRHI_BIN_CAP")) cap = std::min<uint32_t>(cap, std::max<uint32_t>(8u * BIN_PAGE_RECS, ((uint32_t)atoi(getenv("MIRHI_BIN_CAP")) + 511u) & ~511u));
        g.bin_cap = cap; g.sub_cap = g.xcd_bins ? cap / 8u : cap;
        // Pool: the first page of every single-list bin has a fixed place (page = tile); the dynamic part is sized for the
        // (triangle, tile) pairs the scope is likely to produce -- 8 per triangle for small scopes (scattered 50-pixel triangles
        // make 5), towards 1.5 for big meshes (1.2 measured on the 1M-triangle grid) -- plus one partly filled page per list.  A scope
        // that needs more spills into the big list (correct, slower) and the pool is doubled for the next submit.
        const size_t tris = pass.total_tris - pass.first_tri;
        size_t pairs = std::max(std::max(std::min<size_t>(8 * tris, 262144), std::min<size_t>(3 * tris, 786432)), 3 * tris / 2);
        if (pairs > 16 * tris) pairs = 16 * tris;                        // (a binned triangle spans at most 4 x 4 tiles)
        // fixed pages per tile: what the average density fills (x 1.3 for triangles that straddle tiles), at least one, at most eight --
        // a uniform mesh (the 1M-triangle grid: 123 per tile) then bins without a single allocation, a concentrated one (the
        // dancer asset) opens pages where its triangles are
        g.fixed_per_tile = g.xcd_bins ? 0u : (uint32_t)std::min<size_t>(8, std::max<size_t>(1, tiles ? (13 * tris / (10 * tiles) + BIN_PAGE_RECS - 1) / BIN_PAGE_RECS : 1));
        if (getenv("MIRHI_FIXED_PAGES") && !g.xcd_bins) g.fixed_per_tile = (uint32_t)std::min(8, std::max(1, atoi(getenv("MIRHI_FIXED_PAGES"))));   // (tests, A/B runs)
        g.fixed_pages = g.fixed_per_tile * (uint32_t)tiles;
        // dynamic part: the estimated pairs that the fixed pages will not take (they take at most half of it when the triangles sit
        // in a part of the frame), never less than a quarter of the estimate, plus a partly filled page for one tile in four
        const size_t fixed_capacity = (size_t)g.fixed_pages * BIN_PAGE_RECS;
        const size_t dyn_records = std::max(pairs > fixed_capacity / 2 ? pairs - fixed_capacity / 2 : 0, pairs / 4) * cmd->ws.pool_scale;
        size_t pages = g.fixed_pages + ((dyn_records / BIN_PAGE_RECS + tiles * (g.xcd_bins ? 8 : 1) / 4 + 64 + 7) & ~(size_t)7);
        if (getenv("MIRHI_POOL_PAGES")) pages = g.fixed_pages + 8 * (((size_t)atoi(getenv("MIRHI_POOL_PAGES")) + 7) / 8);      // (pool-exhaustion test)
        g.big_cap = pass.total_tris + pass.total_tris / 4 + 1024;
        geo.push_back(g);
        total_draws += pass.draws.size();
        if (tiles > max_tiles) max_tiles = tiles;
        if (pages > max_pages) max_pages = pages;
        if (g.big_cap > max_big) max_big = g.big_cap;
    }
    Workspace& w = cmd->ws;
    mirhi_result r;
    // Buffers only ever grow, and a frame loop's do not: the steady state allocates nothing.  Counters and page table are cleared
    // when they are new (or after a frame that went wrong on the device: Workspace::dirty) -- never otherwise: every kernel leaves
    // them re-armed (raster_body: bin counters, table rows, pool counters, the other parity's big-list counter), and the fixed
    // words of the counter block do not move when the tile count changes (CTR_*).
    if ((r = grow(&w.bin_pool, &w.bin_pool_bytes, (max_pages ? max_pages : 1) * BIN_PAGE_RECS * sizeof(BinRec))) != MIRHI_OK) return r;
    {
        const bool had = w.bin_table != nullptr && w.bin_table_bytes >= (max_tiles ? max_tiles : 1) * BIN_TABLE_ROW * sizeof(uint32_t);
        if ((r = grow(&w.bin_table, &w.bin_table_bytes, (max_tiles ? max_tiles : 1) * BIN_TABLE_ROW * sizeof(uint32_t))) != MIRHI_OK) return r;
        if (!had) w.dirty = true;
    }
    const size_t pool_pages = w.bin_pool_bytes / (BIN_PAGE_RECS * sizeof(BinRec));
    w.xcd_tiles_last = (!geo.empty() && geo.back().xcd_bins) ? geo.back().tiles_x * (geo.back().r1 - geo.back().r0) : 0u;
    if ((r = grow(&w.big_recs, &w.big_recs_bytes, (max_big ? max_big : 1) * sizeof(BigRec))) != MIRHI_OK) return r;
    {
        size_t counter_bytes = w.counters_words * 4;
        bool any_xcd_bins = false;
        for (const Geo& g : geo) any_xcd_bins |= g.xcd_bins;
        // fixed words (big-list and pool counters), then the bin counters: one per tile -- per tile and XCD in scopes with per-XCD bins -- 64 bytes apart
        const size_t want_words = CTR_BINS + (any_xcd_bins ? 8 : 1) * max_tiles * (size_t)BIN_COUNT_STRIDE;
        const bool had = w.counters && counter_bytes >= want_words * 4;
        if ((r = grow(&w.counters, &counter_bytes, want_words * 4)) != MIRHI_OK) return r;
        w.counters_words = counter_bytes / 4;
        if (!had) w.dirty = true;
    }
    if (getenv("MIRHI_ALWAYS_CLEAR")) w.dirty = true;             // (A/B runs: the round-2 behaviour, two memsets per recording)
    if (!w.status_host) {
        HIP_TRY(hipHostMalloc((void**)&w.status_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer((void**)&w.status_dev, w.status_host, 0));
    }
    w.stats_params_valid = false;
    hand_over_status(cmd);          // an earlier submission's status is not lost to the re-arm below (and may ask for a bigger pool: before the flags are cleared)
    w.grow_pool = false; w.replan = false;
    w.status_host[0] = 0; w.status_host[1] = 0; w.status_host[2] = 0; w.status_host[3] = 0;
    w.big_counts = w.counters + CTR_BIG;
    if (!w.dirty && getenv("MIRHI_VERIFY_IDLE")) {
        // Test hook: what the plan relies on instead of clearing -- every kernel leaves the workspace re-armed -- is checked here, on the
        // host: all bin and pool counters zero, the big-list counter of the next parity zero, every page-table entry PAGE_EMPTY.
        HIP_TRY(hipStreamSynchronize(stream));
        std::vector<uint32_t> c(w.counters_words), t(w.bin_table_bytes / 4);
        HIP_TRY(hipMemcpy(c.data(), w.counters, c.size() * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(t.data(), w.bin_table, t.size() * 4, hipMemcpyDeviceToHost));
        size_t bad_c = 0, bad_t = 0;
        for (size_t i = 0; i < c.size(); i++)       // (the busy-tile counters of the last scope are read and re-armed by the next one)
            bad_c += (c[i] != 0u && i != (size_t)(CTR_BIG + (w.parity ^ 1u)) && !(i >= CTR_ACTIVE && i < CTR_ACTIVE + 2u * 8u * 32u)) ? 1 : 0;
        for (uint32_t v : t) bad_t += v != PAGE_EMPTY ? 1 : 0;
        if (bad_c || bad_t) return fail(MIRHI_ERR_DEVICE, "Vulkan error: workspace not idle between frames: %zu counter words, %zu page-table entries left set", bad_c, bad_t);
    }
    if (w.dirty) {
        HIP_TRY(hipMemsetAsync(w.bin_table, 0xFF, w.bin_table_bytes, stream));          // PAGE_EMPTY
        HIP_TRY(hipMemsetAsync(w.counters, 0, w.counters_words * 4, stream));
        dev->foreign_writes++;
        w.parity = 0;
        w.dirty = false;
        cmd->last_stream = stream; cmd->last_native = nullptr; cmd->pending = true;      // (a submit on another stream or queue waits for the clears)
    }

    // vertex pre-pass jobs: one per distinct (vertex range, camera, object, program class) of each scope
    struct HostJob { VsJob j; size_t out_off; };
    std::vector<std::vector<HostJob>> pass_jobs(cmd->passes.size());
    std::vector<std::vector<DrawDesc>> pass_draws(cmd->passes.size());      // the recorded descriptors stay as recorded (plan cache); these get the plan's fields
    size_t vs_bytes_max = 0, jobs_total = 0;
    for (size_t pi = 0; pi < cmd->passes.size(); pi++) {
        RecordedPass& pass = cmd->passes[pi];
        pass_draws[pi] = pass.draws;
        size_t off = 0; uint32_t slots = 0;
        for (size_t di = 0; di < pass.draws.size(); di++) {
            DrawDesc& dd = pass_draws[pi][di];
            dd.vs_words = 0; dd.vs_out = nullptr; dd.vs_attr = nullptr;
            if (dd.program == MIRHI_PROGRAM_TRIANGLE) continue;
            const uint32_t words = dd.program == MIRHI_PROGRAM_MODEL ? 3u : 5u;
            const uint64_t vbb = pass.draw_vb_bytes[di];
            const uint32_t count = vbb >= 48 ? (uint32_t)((vbb - 48) / dd.stride + 1) : 0u;
            size_t found = SIZE_MAX;
            for (size_t j = 0; j < pass_jobs[pi].size(); j++) {
                const VsJob& J = pass_jobs[pi][j].j;
                if (J.vb == dd.vb && J.camera == dd.camera && J.object == dd.object && J.stride == dd.stride && J.words >= words && J.count >= count) { found = j; break; }
            }
            if (found == SIZE_MAX) {
                HostJob hj{};
                hj.j.vb = dd.vb; hj.j.camera = dd.camera; hj.j.object = dd.object; hj.j.stride = dd.stride; hj.j.count = count;
                hj.j.words = words; hj.j.slot_base = slots; hj.out_off = off;
                slots += (count + GEOM_THREADS - 1) / GEOM_THREADS * GEOM_THREADS;
                off += vs_attr_offset(count) + (((size_t)count * (words - 1u) * 16 + 255) & ~(size_t)255);      // clip stream, then the attribute stream
                pass_jobs[pi].push_back(hj);
                found = pass_jobs[pi].size() - 1;
            }
            dd.vs_words = pass_jobs[pi][found].j.words;
            dd.vs_out = (const void*)(uintptr_t)(pass_jobs[pi][found].out_off + 1);   // offset + 1, patched to a pointer below
            dd.vs_attr = (const void*)(uintptr_t)(pass_jobs[pi][found].out_off + vs_attr_offset(pass_jobs[pi][found].j.count) + 1);
        }
        if (off > vs_bytes_max) vs_bytes_max = off;
        jobs_total += pass_jobs[pi].size();
    }
    if ((r = grow(&w.vs_out, &w.vs_out_bytes, vs_bytes_max ? vs_bytes_max : 256)) != MIRHI_OK) return r;
    {
        size_t carry = 0;
        for (auto& pass : cmd->passes)
            if ((pass.carry_in || pass.carry_out) && !pass.depth_t.ptr) {
                const size_t need = (size_t)pass.color_t.width * pass.color_t.height * 4;
                if (need > carry) carry = need;
            }
        if (carry && (r = grow(&w.carry_depth, &w.carry_depth_bytes, carry)) != MIRHI_OK) return r;
    }
    {
        size_t ord = 0;
        for (auto& pass : cmd->passes)
            if (pass_is_ordered(pass)) { const size_t need = (size_t)(pass.total_tris - pass.first_tri) * sizeof(TriRec); if (need > ord) ord = need; }
        if (ord && (r = grow(&w.ordered, &w.ordered_bytes, ord)) != MIRHI_OK) return r;
    }
    size_t flat_tris = 0;
    for (auto& pass : cmd->passes) {
        bool tri_prog = false;
        for (const DrawDesc& dd : pass.draws) tri_prog |= dd.program == MIRHI_PROGRAM_TRIANGLE;
        if (tri_prog && pass.color_t.format == MIRHI_FORMAT_B8G8R8A8_SRGB && pass.total_tris > flat_tris) flat_tris = pass.total_tris;
    }
    if (flat_tris && (r = grow(&w.flat_color, &w.flat_color_bytes, flat_tris * 4)) != MIRHI_OK) return r;
    {
        size_t pd = 0;
        for (auto& pass : cmd->passes)
            if (pass.draws.size() > 1 && (size_t)(pass.total_tris - pass.first_tri) > pd) pd = pass.total_tris - pass.first_tri;
        if (pd && (r = grow(&w.prim_draw, &w.prim_draw_bytes, pd * 4)) != MIRHI_OK) return r;
    }
    // the parameter block: [2 x PassParams per scope][draw descriptors][vertex jobs]
    const size_t params_bytes = cmd->passes.size() * 2 * sizeof(PassParams);
    w.draws_off = (params_bytes + 255) & ~(size_t)255;
    w.jobs_off = (w.draws_off + total_draws * sizeof(DrawDesc) + 255) & ~(size_t)255;
    const size_t block_bytes = w.jobs_off + jobs_total * sizeof(VsJob);
    if ((r = pblock_reserve(w, block_bytes ? block_bytes : 256)) != MIRHI_OK) return r;
    w.params = reinterpret_cast<PassParams*>(w.pblock);
    DrawDesc* const dev_draws = reinterpret_cast<DrawDesc*>(w.pblock + w.draws_off);
    VsJob* const dev_jobs = reinterpret_cast<VsJob*>(w.pblock + w.jobs_off);
    w.pimage.assign(block_bytes, 0);
    w.draws_count = total_draws;

    size_t draws_done = 0, jobs_done = 0;
    bool any_wide_eligible = false;
    cmd->plan.clear(); cmd->plan_programs.clear(); cmd->plan_tris = 0;
    for (size_t pi = 0; pi < cmd->passes.size(); pi++) {
        RecordedPass& pass = cmd->passes[pi];
        std::vector<DrawDesc>& draws = pass_draws[pi];
        const Geo& g = geo[pi];
        const RecordedPass::Target& ci = pass.color_t;
        PassParams P;
        memset(&P, 0, sizeof P);
        P.width = ci.width; P.height = ci.height;
        P.tiles_x = g.tiles_x; P.tiles_y = g.tiles_y; P.tile_row_begin = g.r0; P.tile_row_end = g.r1; P.tile_row_step = g.rstep;
        P.num_draws = (uint32_t)draws.size(); P.total_tris = pass.total_tris;
        P.draws = dev_draws + draws_done;
        depth_key_setup(P, pass);
        if (pass_is_ordered(pass)) {
            P.ordered_recs = w.ordered; P.ordered_first = pass.first_tri; P.ordered_count = pass.total_tris - pass.first_tri;
            P.ord_depth_test = pass.depth_test; P.ord_depth_write = pass.depth_write; P.ord_depth_op = pass.depth_compare;
            memcpy(P.blend, pass.blend, sizeof P.blend);
            P.idflip = 0; P.pred = 0;               // records carry the plain primitive id; the kernel applies the depth state itself
        }
        memcpy(P.clear_color, pass.info.clear_color, sizeof P.clear_color);
        {   // sRGB OETF + UNORM8 of the clear colour, BGRA byte order (swapchain.rs:561-570)
            auto sat = [](float c) { return c > 0.0f ? (c < 1.0f ? c : 1.0f) : 0.0f; };
            auto enc = [&](float c) { c = sat(c); float e = c <= 0.0031308f ? 12.92f * c : 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f; return (uint32_t)std::nearbyint(sat(e) * 255.0f); };
            P.clear_packed = enc(P.clear_color[2]) | (enc(P.clear_color[1]) << 8) | (enc(P.clear_color[0]) << 16) |
                             ((uint32_t)std::nearbyint(sat(P.clear_color[3]) * 255.0f) << 24);
        }
        P.color_load = pass.info.color_load_op == MIRHI_LOAD_OP_LOAD ? 1u : 0u;
        P.color_format = ci.format;
        P.color = const_cast<uint8_t*>(ci.ptr);
        if (pass.depth_t.ptr) {
            P.depth = (float*)const_cast<uint8_t*>(pass.depth_t.ptr);
            P.depth_load = pass.info.depth_load_op == MIRHI_LOAD_OP_LOAD ? 1u : 0u;
            P.depth_store = pass.info.depth_store_op == MIRHI_STORE_OP_STORE ? 1u : 0u;
        } else if (pass.carry_in || pass.carry_out) {
            P.depth = w.carry_depth;               // no depth attachment: the segments of the scope hand depth over here
        }
        if (pass.carry_in) P.depth_load = 1u;
        if (pass.carry_out) P.depth_store = 1u;
        P.prim_out = (uint32_t*)const_cast<uint8_t*>(pass.prim_t.ptr);
        P.bin_pool = w.bin_pool; P.bin_count = w.counters + CTR_BINS; P.bin_cap = g.bin_cap;
        P.bin_table = w.bin_table; P.pool_next = w.counters + CTR_POOL;
        P.pool_dyn_base = g.fixed_pages; P.pool_dyn_pages = (uint32_t)((pool_pages - g.fixed_pages) / 8);      // per XCD
        P.fixed_recs = g.fixed_per_tile * BIN_PAGE_RECS;
        P.big_recs = w.big_recs; P.big_count = w.big_counts; P.big_count_next = w.big_counts + 1; P.big_cap = g.big_cap;
        P.status = w.status_dev;
        P.frag_stats = dev->frag_stats;
        P.first_prim = pass.first_tri;
        P.prim_draw = draws.size() > 1 ? w.prim_draw : nullptr;
        {
            bool tri_prog = false;
            for (const DrawDesc& dd : draws) tri_prog |= dd.program == MIRHI_PROGRAM_TRIANGLE;
            P.flat_color = (tri_prog && ci.format == MIRHI_FORMAT_B8G8R8A8_SRGB) ? w.flat_color : nullptr;
            P.resolve_flat_only = (P.flat_color && !P.depth_load && P.color_format != 2u && !P.prim_out && !(P.depth && P.depth_store)) ? 1u : 0u;
        }
        {
            const RasterMode mode = raster_mode(pass, (size_t)g.tiles_x * (g.r1 - g.r0), w.spread, w.wide);
            P.tp_max_area = mode.tp_max_area;
            P.alpha_scope = pass_is_masked_plain(pass) ? 1u : 0u;
            P.raster_teams = mode.teams;
            P.raster_wide = mode.wide;          // waves per tile of the wide variants: 0 (four waves), 8 or 16
            any_wide_eligible |= mode.wide_eligible;
            P.sub_cap = g.sub_cap;
            P.count_stride = g.xcd_bins ? (uint32_t)max_tiles : 0u;
        }
        P.xcd_swizzle = getenv("MIRHI_XCD_RUN") ? (uint32_t)atoi(getenv("MIRHI_XCD_RUN")) : 1u;
        P.vs_jobs = dev_jobs + jobs_done;
        P.num_vs_jobs = (uint32_t)pass_jobs[pi].size();
        P.vs_total_slots = 0;
        VsJob* const img_jobs = reinterpret_cast<VsJob*>(w.pimage.data() + w.jobs_off) + jobs_done;
        for (size_t j = 0; j < pass_jobs[pi].size(); j++) {
            HostJob& hj = pass_jobs[pi][j];
            hj.j.out = w.vs_out + hj.out_off;
            P.vs_total_slots = hj.j.slot_base + (hj.j.count + GEOM_THREADS - 1) / GEOM_THREADS * GEOM_THREADS;
            img_jobs[j] = hj.j;
        }
        jobs_done += pass_jobs[pi].size();
        for (DrawDesc& dd : draws)
            if (dd.vs_words) { dd.vs_out = w.vs_out + ((size_t)(uintptr_t)dd.vs_out - 1); dd.vs_attr = w.vs_out + ((size_t)(uintptr_t)dd.vs_attr - 1); }
        uint32_t slots = 0;
        for (DrawDesc& dd : draws) { dd.slot_base = slots; slots += (dd.tri_count + GEOM_THREADS - 1) / GEOM_THREADS * GEOM_THREADS; }
        P.total_slots = slots;
        if (!draws.empty()) memcpy(w.pimage.data() + w.draws_off + draws_done * sizeof(DrawDesc), draws.data(), draws.size() * sizeof(DrawDesc));
        draws_done += draws.size();
        cmd->plan.push_back(P);
        uint32_t progs = 0;
        for (const DrawDesc& dd : draws) progs |= dd.program == 0 ? 1u : ((dd.program == MIRHI_PROGRAM_MODEL_PBR || dd.tex_any_mips || dd.tex_srgb) ? 4u : 2u);
        cmd->plan_programs.push_back(progs ? progs : 1u);
        cmd->plan_tris += pass.total_tris - pass.first_tri;
        // the kernels read their parameters from the block: copy 2*pi + parity of scope pi appends large triangles to counter
        // `parity` and re-arms the other one for the scope that follows on this workspace
        for (uint32_t parity = 0; parity < 2; parity++) {
            PassParams Q = P;
            Q.big_count = w.big_counts + parity;
            Q.big_count_next = w.big_counts + (parity ^ 1u);
            Q.active = w.counters + CTR_ACTIVE + parity * 8u * 32u;
            Q.active_prev = w.counters + CTR_ACTIVE + (parity ^ 1u) * 8u * 32u;
            memcpy(w.pimage.data() + (2 * pi + parity) * sizeof(PassParams), &Q, sizeof Q);
        }
    }
    w.wide_eligible = any_wide_eligible;
    if ((r = pblock_commit(w, stream)) != MIRHI_OK) return r;
    if (!w.pblock_direct) { cmd->last_stream = stream; cmd->last_native = nullptr; cmd->pending = true; }     // (the copy is in the lane's stream)
    { std::lock_guard<std::mutex> lk(dev->mu); cmd->planned = cmd->passes; }      // (settle_readers scans `planned` of pending command buffers under this lock)
    cmd->plan_split_rank = dev->split_rank; cmd->plan_split_world = dev->split_world; cmd->plan_split_layout = dev->split_layout;
    cmd->plan_valid = true;
    dev->stats.workspace_bytes = w.bytes();
    return MIRHI_OK;
}

// ------------------------------------------------------------------------------------------------
// submit + fences
// ------------------------------------------------------------------------------------------------
// Statistics pass: device copies of the command buffer's scope parameters (both parities) whose primitive-id image is the
// workspace's own, built the first time a recorded command buffer is submitted with MIRHI_PROFILE_FRAGMENTS.
static mirhi_result stats_params_for(mirhi_cmd* c) {
    Workspace& w = c->ws;
    if (w.stats_params_valid) return MIRHI_OK;
    size_t pixels = 0;
    for (const PassParams& P : c->plan) pixels = std::max(pixels, (size_t)P.width * P.height);
    mirhi_result r;
    if ((r = grow(&w.stats_prim, &w.stats_prim_bytes, (pixels ? pixels : 1) * 4)) != MIRHI_OK) return r;
    if ((r = grow(&w.stats_params, &w.stats_params_bytes, (c->plan.size() ? c->plan.size() : 1) * 2 * sizeof(PassParams))) != MIRHI_OK) return r;
    std::vector<PassParams> copies;
    for (const PassParams& P0 : c->plan)
        for (uint32_t parity = 0; parity < 2; parity++) {
            PassParams P = P0;
            P.big_count = w.big_counts + parity;
            P.big_count_next = w.big_counts + (parity ^ 1u);
            P.active = w.counters + CTR_ACTIVE + parity * 8u * 32u;
            P.active_prev = w.counters + CTR_ACTIVE + (parity ^ 1u) * 8u * 32u;
            P.prim_out = w.stats_prim;
            P.resolve_flat_only = 0u;
            copies.push_back(P);
        }
    if (!copies.empty()) {
        c->dev->foreign_writes++;
        HIP_TRY(hipMemcpyAsync(w.stats_params, copies.data(), copies.size() * sizeof(PassParams), hipMemcpyHostToDevice, c->dev->stream));
        HIP_TRY(hipStreamSynchronize(c->dev->stream));
    }
    w.stats_params_valid = true;
    return MIRHI_OK;
}

static mirhi_result timing_begin(mirhi_device* dev, uint32_t kernel, uint32_t lane, LaunchTiming* t) {
    hipEvent_t ev[2];
    for (hipEvent_t& e : ev) {
        if (!dev->free_events.empty()) { e = dev->free_events.back(); dev->free_events.pop_back(); }
        else HIP_TRY(hipEventCreate(&e));
    }
    t->start = ev[0]; t->stop = ev[1];
    dev->pending.push_back(TimedDispatch{ev[0], ev[1], kernel, lane});
    return MIRHI_OK;
}

// Attachments of a command buffer's recording that are still live images (dev->mu held).
template <typename F>
static void for_each_attachment(mirhi_device* dev, const mirhi_cmd* c, F&& f) {
    for (const RecordedPass& pass : c->planned)
        for (mirhi_image* img : {pass.info.color_image, pass.info.depth_image, pass.info.prim_id_image})
            if (img && std::find(dev->images.begin(), dev->images.end(), img) != dev->images.end()) f(img);
}
// `c` is about to run on `stream` (or, natively, on `nq`): anything that used one of its attachments last somewhere else, and is not known
// to have finished, goes first.  Between two HIP streams that is an event; with an AQL queue on either side the host waits (a frame loop
// never gets here: its fences have told the library that the earlier frame is done).
static mirhi_result order_attachments(mirhi_device* dev, mirhi_cmd* c, hipStream_t stream, NativeQueue* nq = nullptr) {
    hipError_t err = hipSuccess;
    for_each_attachment(dev, c, [&](mirhi_image* img) {
        const bool elsewhere = nq ? (img->last_native != nq && (img->last_native || img->last_stream)) : (img->last_native != nullptr || (img->last_stream && img->last_stream != stream));
        if (elsewhere && err == hipSuccess) {
            if (img->last_native) { if (!drain_native(dev, img->last_native)) err = hipErrorLaunchTimeOut; }
            else {
                bool live = false;
                for (hipStream_t st : dev->lanes) live |= st == img->last_stream;
                for (hipStream_t st : dev->aux_streams) live |= st == img->last_stream;       // (a band exchange wrote the image last)
                if (live && nq) err = hipStreamSynchronize(img->last_stream);
                else if (live) {
                    if (!dev->order_event) err = hipEventCreateWithFlags(&dev->order_event, hipEventDisableTiming);
                    if (err == hipSuccess) err = hipEventRecord(dev->order_event, img->last_stream);
                    if (err == hipSuccess) err = hipStreamWaitEvent(stream, dev->order_event, 0);
                }
            }
        }
        img->last_stream = nq ? nullptr : stream; img->last_native = nq; img->last_cmd = c; img->last_seq = c->submit_seq;
    });
    if (err == hipErrorLaunchTimeOut && dev->native && dev->native->lost.load(std::memory_order_acquire)) return device_lost(dev);
    if (err != hipSuccess) return hip_fail(err, "attachment ordering across queue lanes");
    return MIRHI_OK;
}
// submission `seq` of `c` has finished: its attachments need no ordering any more (unless something newer used them since)
static void release_attachments(mirhi_device* dev, const mirhi_cmd* c, uint64_t seq) {
    for_each_attachment(dev, c, [&](mirhi_image* img) { if (img->last_cmd == c && img->last_seq == seq) { img->last_stream = nullptr; img->last_native = nullptr; img->last_cmd = nullptr; } });
}

static mirhi_result submit_now(mirhi_device* dev, uint32_t cmd_count, mirhi_cmd* const* cmds, mirhi_fence* fence);

static void submit_thread_main(mirhi_device* dev) {
    (void)hipSetDevice(dev->ordinal);
    uint64_t taken = dev->sq_done.load(std::memory_order_acquire);      // (the thread may have been stopped and started again: the ring goes on where it was)
    for (;;) {
        uint32_t idle = 0;
        while (dev->sq_pushed.load(std::memory_order_acquire) == taken) {
            if (dev->sq_stop.load(std::memory_order_acquire)) return;
            cpu_relax();
            if (++idle > 40000u) {                    // ~100 us without work: sleep until the next submit
                std::unique_lock<std::mutex> lk(dev->sq_mu);
                dev->sq_sleeping.store(true, std::memory_order_seq_cst);
                dev->sq_cv.wait(lk, [&] { return dev->sq_pushed.load(std::memory_order_acquire) != taken || dev->sq_stop.load(std::memory_order_acquire); });
                dev->sq_sleeping.store(false, std::memory_order_seq_cst);
                idle = 0;
            }
        }
        mirhi_device::SubmitJob& job = dev->sq_ring[taken % mirhi_device::SQ_SLOTS];
        const mirhi_result rc = submit_now(dev, job.n, job.cmds, job.fence);
        if (rc != MIRHI_OK) {
            // the caller returned long ago: the failure is reported where the submission's completion is asked for
            std::lock_guard<std::mutex> lock(dev->mu);
            if (job.fence) { job.fence->deferred = rc; job.fence->deferred_msg = g_last_error; job.fence->pending = true; }
            else { dev->deferred = rc; dev->deferred_msg = g_last_error; }
        }
        if (job.fence) job.fence->issued.store(true, std::memory_order_release);
        for (uint32_t i = 0; i < job.n; i++) job.cmds[i]->queued.fetch_sub(1, std::memory_order_release);
        taken++;
        dev->sq_done.store(taken, std::memory_order_release);
    }
}

extern "C" mirhi_result mirhi_device_set_submit_thread(mirhi_device* dev, uint32_t enable) {
    NULL_CHECK(dev, "device");
    if ((enable != 0) == dev->sq_on) return MIRHI_OK;
    if (enable) {
        dev->sq_stop.store(false);
        try { dev->sq_thread = std::thread(submit_thread_main, dev); }
        catch (...) { return fail(MIRHI_ERR_DEVICE, "Vulkan error: could not start the submit thread"); }
        dev->sq_on = true;
    } else {
        drain_submits(dev);
        { std::lock_guard<std::mutex> lk(dev->sq_mu); dev->sq_stop.store(true); }
        dev->sq_cv.notify_all();
        if (dev->sq_thread.joinable()) dev->sq_thread.join();
        dev->sq_on = false;
    }
    return MIRHI_OK;
}

extern "C" mirhi_result mirhi_queue_submit(mirhi_device* dev, uint32_t cmd_count, mirhi_cmd* const* cmds, mirhi_fence* fence) {
    NULL_CHECK(dev, "device");
    if (cmd_count) NULL_CHECK(cmds, "cmds");
    for (uint32_t i = 0; i < cmd_count; i++) {
        NULL_CHECK(cmds[i], "cmds[i]");
        if (cmds[i]->dev != dev) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: command buffer belongs to another device");
        if (cmds[i]->state != CMD_EXECUTABLE) return fail(MIRHI_ERR_DEVICE, "Vulkan error: command buffer %u is not in the executable state (call end())", i);
    }
    if (fence && fence->dev != dev) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: fence belongs to another device");
    // a lost device refuses a submit here, in the caller's thread, whether or not a submit thread would have issued it (vkQueueSubmit: VK_ERROR_DEVICE_LOST)
    if (dev->native && dev->native->lost.load(std::memory_order_acquire)) return device_lost(dev);
    constexpr uint32_t JOB_CMDS = sizeof(mirhi_device::SubmitJob::cmds) / sizeof(mirhi_cmd*);
    if (!dev->sq_on || cmd_count > JOB_CMDS) {
        drain_submits(dev);
        return submit_now(dev, cmd_count, cmds, fence);
    }
    // Submit thread on: the job is queued, the launches happen there.  What the caller may look at next is set here, in its own thread.
    std::lock_guard<std::mutex> producer(dev->sq_push_mu);
    const uint64_t slot = dev->sq_pushed.load(std::memory_order_relaxed);
    while (slot - dev->sq_done.load(std::memory_order_acquire) >= mirhi_device::SQ_SLOTS) cpu_relax();
    mirhi_device::SubmitJob& job = dev->sq_ring[slot % mirhi_device::SQ_SLOTS];
    job.n = cmd_count; job.fence = fence;
    for (uint32_t i = 0; i < cmd_count; i++) { job.cmds[i] = cmds[i]; cmds[i]->queued.fetch_add(1, std::memory_order_relaxed); }
    if (fence) { fence->issued.store(false, std::memory_order_relaxed); fence->pending = true; fence->signaled = false; }
    dev->sq_pushed.store(slot + 1, std::memory_order_seq_cst);
    if (dev->sq_sleeping.load(std::memory_order_seq_cst)) { std::lock_guard<std::mutex> lk(dev->sq_mu); dev->sq_cv.notify_one(); }
    return MIRHI_OK;
}

static mirhi_result submit_now(mirhi_device* dev, uint32_t cmd_count, mirhi_cmd* const* cmds, mirhi_fence* fence) {
    HIP_TRY(hipSetDevice(dev->ordinal));
    for (uint32_t i = 0; i < cmd_count; i++)
        if ((cmds[i]->ws.grow_pool && !getenv("MIRHI_POOL_PAGES")) || cmds[i]->ws.replan) {   // an earlier submission ran out of bin pages (a bigger pool, the same plan) or showed a spread-out mesh (one team)
            const mirhi_result rp = build_plan(cmds[i], true);
            if (rp != MIRHI_OK) return rp;
        }
    // dev->mu guards the bookkeeping (unchecked list, statistics, attachment ordering, timed-dispatch events), not the launches: with
    // the submit thread on, the render thread must be able to complete a fence while this thread is inside hipLaunchKernel
    std::unique_lock<std::mutex> lock(dev->mu);
    bool keep_locked = dev->profiling != 0;
    // Batched form: the command buffers of one submit, when each is one plain rendering scope of the same shape and kernel variants
    // (the frames of a frame loop), share one vertex, one geometry and one raster launch on the first one's queue lane -- the
    // ramp-up and drain of a kernel and the latency chain of the geometry kernel are paid once per batch, not once per frame.
    // Only frames that are independent of each other may share a launch: every scope clears (no LOAD of colour or depth -- what it
    // would load might be written by another scope of the batch), no two scopes share a colour, depth or primitive-id attachment, and
    // all command buffers sit on the first one's queue lane (so the batch keeps their order against earlier work of that lane).
    // Anything else runs command buffer by command buffer, in submission order on each lane.
    bool batched = cmd_count >= 2 && cmd_count <= (uint32_t)MAX_BATCH && dev->profiling == 0 && !native_env().no_batch;
    for (uint32_t i = 0; batched && i < cmd_count; i++) {
        const mirhi_cmd* c = cmds[i];
        batched = c->plan.size() == 1 && raster_batchable(c->plan[0]) && c->lane == cmds[0]->lane &&
                  raster_variant_key(c->plan[0], c->plan_programs[0]) == raster_variant_key(cmds[0]->plan[0], cmds[0]->plan_programs[0]) &&
                  c->plan_programs[0] == cmds[0]->plan_programs[0] && !c->plan[0].color_load && !c->plan[0].depth_load;
        for (uint32_t j = 0; batched && j < i; j++) {
            const PassParams& A = cmds[j]->plan[0]; const PassParams& B = c->plan[0];
            batched = cmds[j] != c && A.color != B.color && !(A.depth && A.depth == B.depth) && !(A.prim_out && A.prim_out == B.prim_out);
        }
    }
    // The fence rides on the submit's last dispatch (its completion signal: hipExtLaunchKernelGGL stop event) when there is one and
    // everything of the submit runs on one stream -- an event RECORD is a command of its own in the stream: 4.5 us of stream time and
    // a round trip of 12-14 us against 6-9 us (tools/microbench/fence_latency.hip).
    // Native dispatch (mirhi_native.h): the submit's kernels go out as AQL packets on the lane's own queue -- when nothing of the submit
    // needs the HIP stream: no timed dispatches, no batch, no tile split (the band exchange lives on HIP streams), no ordered segment
    // (its clear is a HIP memset), every command buffer on one lane.
    // A device made on the caller's stream (mirhi_device_create_on_stream) promised that lane 0's work is issued on that stream: submits to lane 0 stay in
    // stream order (HIP launches) unless the caller opted in (mirhi_device_set_native_dispatch); lanes the library made itself are the library's.
    bool use_native = !batched && cmd_count >= 1 && dev->native && dev->native->ok && dev->profiling == 0 &&
                      (dev->owns_stream || dev->native_on_external || cmds[0]->lane != 0u);
    for (uint32_t i = 0; use_native && i < cmd_count; i++) {
        use_native = cmds[i]->lane == cmds[0]->lane && cmds[i]->lane < dev->lanes.size();
        for (const PassParams& P : cmds[i]->plan) use_native = use_native && !P.ordered_recs;
    }
    NativeQueue* nq = use_native ? native_lane(dev, cmds[0]->lane) : nullptr;
    use_native = nq != nullptr;
    if (use_native) keep_locked = true;       // (an AQL queue has one producer at a time, and a dispatch takes 0.2 us: the device lock stays held)
    hipEvent_t fence_stop = nullptr;
    bool fence_attached = false;
    if (fence) fence->native_wait = false;
    if (use_native && dev->native->lost.load(std::memory_order_acquire)) return device_lost(dev);
    if (fence && use_native) {
        fence->native_q = nq;
        if (!fence->native_sig.handle && hsa_signal_create(0, 0, nullptr, &fence->native_sig) != HSA_STATUS_SUCCESS) return fail(MIRHI_ERR_DEVICE, "Vulkan error: hsa_signal_create for a fence failed");
        hsa_signal_store_relaxed(fence->native_sig, 1);
    }
    if (fence && !use_native) {
        if (!fence->event) HIP_TRY(hipEventCreate(&fence->event));
        bool one_stream = cmd_count >= 1 && dev->profiling == 0 && !native_env().fence_record;
        for (uint32_t i = 1; one_stream && i < cmd_count && !batched; i++) one_stream = cmds[i]->lane == cmds[0]->lane;
        if (one_stream) {
            const mirhi_cmd* last = cmds[cmd_count - 1];
            const bool has_launch = !last->plan.empty() && last->plan.back().tile_row_end > last->plan.back().tile_row_begin && last->plan.back().tiles_x != 0u;
            if (has_launch) fence_stop = fence->event;
        }
    }
    if (batched) {
        hipStream_t stream = dev->lanes[cmds[0]->lane < dev->lanes.size() ? cmds[0]->lane : 0];
        const PassParams* P[MAX_BATCH]; const PassParams* dp[MAX_BATCH]; uint32_t* big[MAX_BATCH];
        for (uint32_t i = 0; i < cmd_count; i++) {
            mirhi_cmd* c = cmds[i];
            if (c->pending && c->last_stream && c->last_stream != stream) HIP_TRY(hipStreamSynchronize(c->last_stream));   // (its workspace may still be in use there)
            if (c->pending && c->last_native && !drain_native(dev, c->last_native)) return device_lost(dev);
            c->last_stream = stream; c->last_native = nullptr; c->pending = true; c->submit_seq++;
            { const mirhi_result ro = order_attachments(dev, c, stream); if (ro != MIRHI_OK) return ro; }
            if (std::find(dev->unchecked.begin(), dev->unchecked.end(), c) == dev->unchecked.end()) dev->unchecked.push_back(c);
            P[i] = &c->plan[0];
            dp[i] = c->ws.params + c->ws.parity;
            big[i] = c->ws.big_counts + c->ws.parity;
            c->ws.parity ^= 1u;
            dev->stats.frames_submitted++;
            dev->stats.triangles_submitted += c->plan[0].total_tris;
        }
        if (!keep_locked) lock.unlock();
        hipError_t le = launch_vertex_batch(P, dp, cmd_count, stream);
        if (le == hipSuccess) le = launch_geometry_batch(P, dp, cmd_count, stream);
        if (le == hipSuccess) le = launch_raster_batch(P, dp, big, cmd_count, cmds[0]->plan_programs[0], stream, fence_stop);
        if (!keep_locked) lock.lock();
        HIP_TRY(le);
        fence_attached = fence_stop != nullptr;
    }
    for (uint32_t i = 0; !batched && i < cmd_count; i++) {
        mirhi_cmd* c = cmds[i];
        hipStream_t stream = dev->lanes[c->lane < dev->lanes.size() ? c->lane : 0];
        if (c->pending && !use_native && c->last_native && !drain_native(dev, c->last_native)) return device_lost(dev);
        if (c->pending && c->last_stream && (use_native || c->last_stream != stream)) HIP_TRY(hipStreamSynchronize(c->last_stream));
        if (c->pending && use_native && c->last_native && c->last_native != nq && !drain_native(dev, c->last_native)) return device_lost(dev);
        // Frames in flight, this one included (command buffers submitted and not yet known to have finished).  The wide mesh variants trade
        // throughput for latency -- a frame alone on the chip finishes sooner (C3 raster 32 -> 25 us), four frames in flight leave each
        // other less room (C3 16.2 -> 19.8 us per frame) -- so a submit takes them only while the queue is shallow: the reference's
        // MAX_FRAMES_IN_FLIGHT = 2 loop does, a loop that keeps four frames queued gets the plain / two-team variants.
        int in_flight = 1;
        for (const mirhi_cmd* o : dev->cmds) in_flight += (o != c && o->pending) ? 1 : 0;
        const bool allow_wide = in_flight <= 2 || getenv("MIRHI_RASTER_WIDE") != nullptr;
        c->last_stream = use_native ? nullptr : stream; c->last_native = use_native ? nq : nullptr; c->pending = true; c->submit_seq++;
        { const mirhi_result ro = order_attachments(dev, c, stream, use_native ? nq : nullptr); if (ro != MIRHI_OK) return ro; }
        if (std::find(dev->unchecked.begin(), dev->unchecked.end(), c) == dev->unchecked.end()) dev->unchecked.push_back(c);
        for (size_t pi = 0; pi < c->plan.size(); pi++) {
            const PassParams& P = c->plan[pi];
            // alternate the big-list counter: the raster kernel zeroes the other one for the next scope
            const PassParams* dp = c->ws.params + 2 * pi + c->ws.parity;
            uint32_t* big_count = c->ws.big_counts + c->ws.parity;
            c->ws.parity ^= 1u;
            // ordered segment: slots of primitives that no draw of the segment covers (a Never draw keeps its ids) must read
            // as "no coverage" -- an all-zero record is a degenerate triangle whose edge functions are negative everywhere
            if (P.ordered_recs && P.ordered_count) HIP_TRY(hipMemsetAsync(P.ordered_recs, 0, (size_t)P.ordered_count * sizeof(TriRec), stream));
            LaunchTiming tv{}, tg{}, tc{}, tr{};
            if (use_native) {
                tv.native = nq; tg.native = nq; tr.native = nq;
                // the scope's first kernels see what the host wrote (parameter block, buffers uploaded since); its raster kernel publishes the frame
                // -- when something other than this library's kernels wrote device memory since this queue last acquired at system scope
                const int scope_mode = native_env().system_scope;   // 1: system scope on every packet, 2: on every scope's first (A/B runs)
                const bool sys = scope_mode == 1;
                const uint64_t foreign = dev->foreign_writes.load(std::memory_order_acquire);
                const bool head_sys = sys || scope_mode == 2 || nq->seen_foreign != foreign || c->ws.foreign;
                nq->seen_foreign = foreign; c->ws.foreign = false;
                tv.native_flags = (head_sys ? NATIVE_ACQUIRE_SYSTEM : 0u) | (sys ? NATIVE_RELEASE_SYSTEM : 0u);
                tg.native_flags = (((P.vs_total_slots == 0u && head_sys) || sys) ? NATIVE_ACQUIRE_SYSTEM : 0u) | (sys ? NATIVE_RELEASE_SYSTEM : 0u);      // (behind a vertex kernel: that one took the acquire)
                tr.native_flags = NATIVE_RELEASE_SYSTEM | (sys ? NATIVE_ACQUIRE_SYSTEM : 0u);
            }
            {   // small scopes: fewer triangles per geometry wave (GeometryHead::tris_per_wave) -- the chip is mostly idle, a shorter wave is a shorter frame
                const uint32_t geo_waves = P.total_slots / (uint32_t)GEOM_THREADS;
                tg.tris_per_wave = native_env().geom_tpw ? (uint32_t)native_env().geom_tpw : (geo_waves <= 256u ? 16u : (geo_waves <= 512u ? 32u : 64u));
                if (pi < c->planned.size() && c->planned[pi].draws.size() == 1u) tg.head_draw = &c->planned[pi].draws[0];      // (GeometryHead::vb0: launch_geometry decides)
            }
            // (timing may be restricted to one queue lane -- bits 8..15 of the mask hold lane + 1 -- so that the other lanes run
            // untimed: a timed dispatch completes through its own signal and does not overlap its neighbours the way an untimed one does)
            const uint32_t only_lane = (dev->profiling >> 8) & 0xFFu;
            const bool timed = (dev->profiling & MIRHI_PROFILE_TIMING) != 0 && (only_lane == 0u || only_lane - 1u == c->lane);
            const bool counted = (dev->profiling & MIRHI_PROFILE_FRAGMENTS) != 0;
            if (timed) {
                mirhi_result r;
                if (P.vs_total_slots && (r = timing_begin(dev, MIRHI_KERNEL_VERTEX, c->lane, &tv)) != MIRHI_OK) return r;
                if (P.total_slots && (r = timing_begin(dev, MIRHI_KERNEL_GEOMETRY, c->lane, &tg)) != MIRHI_OK) return r;
            }
            if (!keep_locked) lock.unlock();
            hipError_t le = launch_vertex(P, dp, stream, tv);
            if (le == hipSuccess) le = launch_geometry(P, dp, stream, tg);
            if (!keep_locked) lock.lock();
            HIP_TRY(le);
            const bool has_tiles = P.tile_row_end > P.tile_row_begin && P.tiles_x;
            const uint32_t* winners = nullptr;
            if (counted && has_tiles && !P.ordered_recs) {          // (ordered -- blended -- segments are not counted)
                mirhi_result r;
                if (timed && (r = timing_begin(dev, MIRHI_KERNEL_FRAGMENT_COUNT, c->lane, &tc)) != MIRHI_OK) return r;
                HIP_TRY(launch_fragment_count(P, dp, big_count, stream, tc));
                dev->frag_scopes++;
                // winners are counted from a primitive-id image: the scope's own if every pixel of it is written (no LOAD),
                // else the workspace's, cleared to NO_PRIM, through a copy of the parameters -- the raster kernels know nothing
                // of the statistics
                if (P.prim_out && !P.color_load) winners = P.prim_out;
                else {
                    if ((r = stats_params_for(c)) != MIRHI_OK) return r;
                    HIP_TRY(hipMemsetAsync(c->ws.stats_prim, 0xFF, (size_t)P.width * P.height * 4, stream));
                    dp = c->ws.stats_params + (dp - c->ws.params);
                    winners = c->ws.stats_prim;
                }
            }
            if (timed && has_tiles) { mirhi_result r = timing_begin(dev, MIRHI_KERNEL_RASTER, c->lane, &tr); if (r != MIRHI_OK) return r; }
            if (fence_stop && i + 1 == cmd_count && pi + 1 == c->plan.size() && has_tiles) { tr.stop = fence_stop; fence_attached = true; }   // (never together with `timed`)
            if (fence && use_native && i + 1 == cmd_count && pi + 1 == c->plan.size() && has_tiles) { tr.native_signal = fence->native_sig.handle; fence_attached = true; }
            if (!keep_locked) lock.unlock();
            le = launch_raster(P, dp, big_count, c->plan_programs[pi], stream, tr, allow_wide);
            if (!keep_locked) lock.lock();
            HIP_TRY(le);
            if (winners) HIP_TRY(launch_winner_count(winners, P.width * P.height, dev->frag_stats, stream));
            dev->stats.frames_submitted++;
            dev->stats.triangles_submitted += P.total_tris;
        }
    }
    if (fence && use_native) {
        if (!fence_attached) {            // nothing of the submit carried the signal (no tiles to raster): wait here, the fence is signalled at once
            native_queue_drain(nq);
            hsa_signal_store_relaxed(fence->native_sig, 0);
        }
        fence->native_wait = true;
        fence->pending = true; fence->signaled = false;
        fence->cmds.assign(cmds, cmds + cmd_count);
        fence->seqs.resize(cmd_count);
        for (uint32_t i = 0; i < cmd_count; i++) fence->seqs[i] = cmds[i]->submit_seq;
        fence->deferred = MIRHI_OK; fence->deferred_msg.clear();
    } else if (fence) {
        if (!fence_attached) {
            // the fence follows the last command buffer's lane and waits for the other lanes used by this submit
            hipStream_t fstream = cmd_count ? cmds[cmd_count - 1]->last_stream : dev->stream;
            for (uint32_t i = 0; i + 1 < cmd_count; i++) {
                hipStream_t other = cmds[i]->last_stream;
                if (other != fstream) {
                    if (!fence->join) HIP_TRY(hipEventCreateWithFlags(&fence->join, hipEventDisableTiming));
                    HIP_TRY(hipEventRecord(fence->join, other));
                    HIP_TRY(hipStreamWaitEvent(fstream, fence->join, 0));
                }
            }
            HIP_TRY(hipEventRecord(fence->event, fstream));
        }
        fence->pending = true; fence->signaled = false;
        fence->cmds.assign(cmds, cmds + cmd_count);
        fence->seqs.resize(cmd_count);
        for (uint32_t i = 0; i < cmd_count; i++) fence->seqs[i] = cmds[i]->submit_seq;
        fence->deferred = MIRHI_OK; fence->deferred_msg.clear();
    }
    return MIRHI_OK;
}

extern "C" mirhi_result mirhi_fence_create(mirhi_device* dev, uint32_t signaled, mirhi_fence** out) {
    NULL_CHECK(dev, "device"); NULL_CHECK(out, "out");
    mirhi_fence* f = new (std::nothrow) mirhi_fence();
    if (!f) return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: host allocation failed");
    f->dev = dev; f->signaled = signaled != 0;
    { std::lock_guard<std::mutex> lock(dev->mu); dev->fences.push_back(f); }
    dev->children++;
    *out = f;
    return MIRHI_OK;
}
// device-side status of one finished command buffer -> stats + error (the same text whether a fence or wait_idle finds it)
static mirhi_result status_of(mirhi_device* dev, mirhi_cmd* c) {
    mirhi_result r = MIRHI_OK;
    if (c->ws.status_host) {
        dev->stats.last_status = c->ws.status_host[0];
        dev->stats.last_big_list = c->ws.status_host[1];
        dev->stats.last_bin_pages = c->ws.status_host[2];
        // (the plan feedback below rewrites flags build_plan reads: not while the submit thread may be re-planning a queued submission of this command
        // buffer -- the status words stay set and the next look at them, with nothing queued, acts on them)
        const bool feedback = c->queued.load(std::memory_order_acquire) == 0;
        if (feedback && (c->ws.status_host[0] & STATUS_POOL_EXHAUSTED) && !c->ws.grow_pool) {
            // not an error: the records went to the big list and the frame is complete; the pool is doubled in front of the next submit
            c->ws.grow_pool = true;
            if (c->ws.pool_scale < 64u) c->ws.pool_scale *= 2u;
        }
        if (feedback && (c->ws.status_host[3] & 0x80000000u)) {
            // busy tiles of the scope before this one: few of them = a mesh in a part of the frame = the wide variant (and back), see Workspace::wide
            // Sixteen waves per tile are one workgroup per CU at a time: for up to ~240 busy tiles (the dancer asset: 232; raster 39 -> 30 us);
            // eight waves are two per CU: up to ~512 (the 70k-triangle sphere: 419; 32.5 -> 24.8 us, sixteen: 31.7 in two rounds).  Hysteresis
            // of a quarter so that a frame loop does not flip between two plans.
            c->ws.busy_tiles = c->ws.status_host[3] & 0x7FFFFFFFu; c->ws.busy_known = true;
            const uint32_t b = c->ws.busy_tiles, cur = c->ws.wide;
            uint32_t want = cur;
            if (b == 0u) want = 0u;
            else if (b <= (cur == 16u ? 300u : 240u)) want = 16u;
            else if (b <= (cur == 8u ? 640u : 512u)) want = 8u;
            else want = 0u;
            if (want != cur && c->ws.wide_eligible && !getenv("MIRHI_RASTER_WIDE")) { c->ws.wide = want; c->ws.replan = true; }
        }
        if (feedback && !c->ws.spread && c->ws.xcd_tiles_last && c->ws.status_host[2] > 2u * c->ws.xcd_tiles_last) { c->ws.spread = true; c->ws.replan = true; c->ws.spread_tris = c->plan_tris; }
        if (c->ws.status_host[0] & (STATUS_PAGE_TIMEOUT | STATUS_BIG_OVERFLOW)) c->ws.dirty = true;     // counters / page table are cleared before the next frame
        if (c->ws.status_host[0] & STATUS_PAGE_TIMEOUT)
            r = fail(MIRHI_ERR_DEVICE, "Vulkan error: rasterizer bin page was never published; frame is incomplete");
        if (c->ws.status_host[0] & STATUS_ALPHA_TEST_TEXTURED)
            r = fail(MIRHI_ERR_PIPELINE, "Pipeline error: MODEL_PBR alpha cutoff together with a base colour texture needs a per-fragment discard: set mirhi_pipeline_desc.fragment_discard_enable on that pipeline; the draw was skipped");
        if (c->ws.status_host[0] & STATUS_BIG_OVERFLOW)
            r = fail(MIRHI_ERR_DEVICE, "Vulkan error: rasterizer large-triangle list overflowed (%u entries); frame is incomplete", c->ws.status_host[1]);
    }
    return r;
}
static mirhi_result check_status_words(mirhi_device* dev) {
    std::lock_guard<std::mutex> lock(dev->mu);
    mirhi_result r = MIRHI_OK;
    for (mirhi_cmd* c : dev->unchecked) { const mirhi_result rc = status_of(dev, c); if (rc != MIRHI_OK) r = rc; }
    dev->unchecked.clear();
    for (mirhi_cmd* c : dev->cmds) cmd_finished(c);       // (called with every lane synchronised)
    for (mirhi_image* img : dev->images) { img->last_stream = nullptr; img->last_native = nullptr; img->last_cmd = nullptr; }
    if (dev->deferred != MIRHI_OK) { if (r == MIRHI_OK) { r = dev->deferred; g_last_error = dev->deferred_msg; } dev->deferred = MIRHI_OK; dev->deferred_msg.clear(); }
    return r;
}
static mirhi_result fence_complete(mirhi_fence* f) {
    f->pending = false; f->signaled = true;
    mirhi_result r = MIRHI_OK;
    {
        std::lock_guard<std::mutex> lock(f->dev->mu);          // what a fence has reported is not reported again by wait_idle
        for (size_t i = 0; i < f->cmds.size(); i++) {
            mirhi_cmd* c = f->cmds[i];
            const mirhi_result rc = status_of(f->dev, c); if (rc != MIRHI_OK) r = rc;
            release_attachments(f->dev, c, f->seqs[i]);
            if (c->submit_seq == f->seqs[i]) {              // (a later submission of the same command buffer is still its own fence's business)
                cmd_finished(c);
                auto& u = f->dev->unchecked;
                u.erase(std::remove(u.begin(), u.end(), c), u.end());
            }
        }
    }
    f->cmds.clear(); f->seqs.clear();
    if (f->deferred != MIRHI_OK) { if (r == MIRHI_OK) { r = f->deferred; g_last_error = f->deferred_msg; } f->deferred = MIRHI_OK; f->deferred_msg.clear(); }
    return r;
}
extern "C" mirhi_result mirhi_fence_wait(mirhi_fence* f, uint64_t timeout_ns) {
    NULL_CHECK(f, "fence");
    if (f->signaled) return MIRHI_OK;
    if (f->pending && !f->issued.load(std::memory_order_acquire)) {
        // its submission still waits in the submit thread's queue (microseconds): the event is recorded when it is issued
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::nanoseconds(timeout_ns < 4000000000ull ? timeout_ns : 4000000000ull);
        for (uint32_t it = 0; !f->issued.load(std::memory_order_acquire); it++) {
            cpu_relax();
            if (timeout_ns != UINT64_MAX && (it & 255u) == 255u && std::chrono::steady_clock::now() >= deadline) return fail(MIRHI_TIMEOUT, "Vulkan error: TIMEOUT");
        }
    }
    if (!f->pending) {
        // unsignaled and nothing submitted: Vulkan would block until the timeout
        if (timeout_ns == UINT64_MAX) return fail(MIRHI_ERR_DEVICE, "Vulkan error: waiting forever on a fence that was never submitted");
        std::this_thread::sleep_for(std::chrono::nanoseconds(timeout_ns < 1000000000ull ? timeout_ns : 1000000000ull));
        return fail(MIRHI_TIMEOUT, "Vulkan error: TIMEOUT");
    }
    if (f->native_wait) {
        // native dispatch: the completion signal of the submission's last packet, a word in host memory (1 -> 0).  Bounded whatever the caller's timeout:
        // a queue that stops making progress for the device's deadline is a lost device (VK_ERROR_DEVICE_LOST), not a thread that spins forever.
        const int w = native_signal_wait(f->native_sig, timeout_ns, f->dev->native, f->native_q, "waiting for a fence");
        if (w == 1) return fail(MIRHI_TIMEOUT, "Vulkan error: TIMEOUT");
        if (w == 2) return device_lost(f->dev);
        f->native_wait = false;
        return fence_complete(f);
    }
    if (!f->event) {
        // nothing was ever recorded for this submission: it failed in the submit thread before its first launch (the error is the deferred one)
        f->pending = false;
        if (f->deferred != MIRHI_OK) { const mirhi_result r = f->deferred; g_last_error = f->deferred_msg; f->deferred = MIRHI_OK; f->deferred_msg.clear(); f->cmds.clear(); f->seqs.clear(); return r; }
        return fail(MIRHI_ERR_DEVICE, "Vulkan error: fence has no submission to wait for");
    }
    HIP_TRY(hipSetDevice(f->dev->ordinal));
    if (timeout_ns == UINT64_MAX) {
        // a frame loop's fence is due within microseconds: poll (60 ns per query) for a while before blocking in the runtime
        const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
        for (uint32_t it = 0;; it++) {
            hipError_t e = hipEventQuery(f->event);
            if (e == hipSuccess) return fence_complete(f);
            if (e != hipErrorNotReady) return hip_fail(e, "hipEventQuery");
            (void)hipGetLastError();
            // (a query takes the runtime's locks: back to back from this thread it starves the thread that is launching)
            for (int k = 0; k < 16; k++) cpu_relax();
            if ((it & 63u) == 63u && std::chrono::steady_clock::now() >= spin_until) break;
        }
        HIP_TRY(hipEventSynchronize(f->event));
        return fence_complete(f);
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::nanoseconds(timeout_ns);
    for (;;) {
        hipError_t e = hipEventQuery(f->event);
        if (e == hipSuccess) return fence_complete(f);
        if (e != hipErrorNotReady) return hip_fail(e, "hipEventQuery");
        (void)hipGetLastError();
        if (std::chrono::steady_clock::now() >= deadline) return fail(MIRHI_TIMEOUT, "Vulkan error: TIMEOUT");
        std::this_thread::yield();
    }
}
extern "C" mirhi_result mirhi_fence_reset(mirhi_fence* f) {
    NULL_CHECK(f, "fence");
    if (f->pending) {   // resetting a fence that is still in flight is invalid in Vulkan; drain it first
        while (!f->issued.load(std::memory_order_acquire)) cpu_relax();
        HIP_TRY(hipSetDevice(f->dev->ordinal));
        if (f->native_wait) { (void)native_signal_wait(f->native_sig, UINT64_MAX, f->dev->native, f->native_q, "resetting a pending fence"); f->native_wait = false; }
        else if (f->event) HIP_TRY(hipEventSynchronize(f->event));
        (void)fence_complete(f);
    }
    f->signaled = false;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_fence_status(mirhi_fence* f) {
    NULL_CHECK(f, "fence");
    if (f->signaled) return MIRHI_OK;
    if (!f->pending || !f->issued.load(std::memory_order_acquire)) return MIRHI_NOT_READY;
    if (f->native_wait) {
        if (hsa_signal_load_scacquire(f->native_sig) != 0) return MIRHI_NOT_READY;
        f->native_wait = false; (void)fence_complete(f);
        return MIRHI_OK;
    }
    if (!f->event) return MIRHI_NOT_READY;        // (a submission that failed in the submit thread: mirhi_fence_wait reports it)
    (void)hipSetDevice(f->dev->ordinal);
    hipError_t e = hipEventQuery(f->event);
    if (e == hipSuccess) { (void)fence_complete(f); return MIRHI_OK; }
    (void)hipGetLastError();
    return MIRHI_NOT_READY;
}
extern "C" mirhi_result mirhi_fence_destroy(mirhi_fence* f) {
    NULL_CHECK(f, "fence");
    (void)hipSetDevice(f->dev->ordinal);
    while (!f->issued.load(std::memory_order_acquire)) cpu_relax();
    if (f->pending && f->native_wait) (void)native_signal_wait(f->native_sig, UINT64_MAX, f->dev->native, f->native_q, "destroying a pending fence");
    else if (f->pending && f->event) (void)hipEventSynchronize(f->event);
    if (f->native_sig.handle) (void)hsa_signal_destroy(f->native_sig);
    if (f->event) (void)hipEventDestroy(f->event);
    if (f->join) (void)hipEventDestroy(f->join);
    { std::lock_guard<std::mutex> lock(f->dev->mu); auto& v = f->dev->fences; v.erase(std::remove(v.begin(), v.end(), f), v.end()); }
    f->dev->children--;
    delete f;
    return MIRHI_OK;
}

// ------------------------------------------------------------------------------------------------
// measurement
// ------------------------------------------------------------------------------------------------
extern "C" mirhi_result mirhi_device_set_profiling(mirhi_device* dev, uint32_t enable) {
    NULL_CHECK(dev, "device");
    drain_submits(dev);
    std::lock_guard<std::mutex> lock(dev->mu);
    dev->profiling = enable & (MIRHI_PROFILE_TIMING | MIRHI_PROFILE_FRAGMENTS | 0xFF00u);
    return MIRHI_OK;
}
// Reads the timed dispatches back: duration = elapsed(start, stop) of the dispatch's own pair; position on the timeline =
// distance of its end from the end of the first dispatch since the last reset (hipEventElapsedTime between the stop events of
// two dispatches is the difference of their completion timestamps, whichever streams they ran on).
static mirhi_result drain_events(mirhi_device* dev) {
    HIP_TRY(hipSetDevice(dev->ordinal));
    for (TimedDispatch& p : dev->pending) {
        HIP_TRY(hipEventSynchronize(p.stop));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, p.start, p.stop));
        bool keep_stop = false;
        double end_ms;
        if (!dev->base_stop) { dev->base_stop = p.stop; dev->base_end_ms = ms; end_ms = ms; keep_stop = true; }
        else { float rel = 0.0f; HIP_TRY(hipEventElapsedTime(&rel, dev->base_stop, p.stop)); end_ms = dev->base_end_ms + rel; }
        if (p.kernel < MIRHI_KERNEL_COUNT) { dev->total_ms[p.kernel] += ms; dev->launches[p.kernel]++; }
        if (dev->timeline.size() < (1u << 20)) dev->timeline.push_back(mirhi_dispatch_time{p.kernel, p.lane, (end_ms - ms) * 1e3, end_ms * 1e3});
        dev->free_events.push_back(p.start);
        if (!keep_stop) dev->free_events.push_back(p.stop);
    }
    dev->pending.clear();
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_kernel_time(mirhi_device* dev, mirhi_kernel_id kernel, double* total_ms, uint64_t* launches) {
    NULL_CHECK(dev, "device");
    if ((int)kernel < 0 || (int)kernel >= MIRHI_KERNEL_COUNT) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: unknown kernel id %d", (int)kernel);
    drain_submits(dev);
    std::lock_guard<std::mutex> lock(dev->mu);
    mirhi_result r = drain_events(dev);
    if (r != MIRHI_OK) return r;
    if (total_ms) *total_ms = dev->total_ms[kernel];
    if (launches) *launches = dev->launches[kernel];
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_timeline(mirhi_device* dev, mirhi_dispatch_time* out, uint32_t capacity, uint32_t* count) {
    NULL_CHECK(dev, "device"); NULL_CHECK(count, "count");
    drain_submits(dev);
    std::lock_guard<std::mutex> lock(dev->mu);
    mirhi_result r = drain_events(dev);
    if (r != MIRHI_OK) return r;
    *count = (uint32_t)dev->timeline.size();
    if (out) for (uint32_t i = 0; i < capacity && i < *count; i++) out[i] = dev->timeline[i];
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_fragment_stats(mirhi_device* dev, uint64_t* shaded_pixels, uint64_t* covered_fragments, uint64_t* scopes) {
    NULL_CHECK(dev, "device");
    { mirhi_result r0 = sync_all_lanes(dev); if (r0 != MIRHI_OK) return r0; }
    unsigned long long v[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(v, dev->frag_stats, sizeof v, hipMemcpyDeviceToHost, dev->stream));
    HIP_TRY(hipStreamSynchronize(dev->stream));
    if (shaded_pixels) *shaded_pixels = v[0];
    if (covered_fragments) *covered_fragments = v[1];
    if (scopes) { std::lock_guard<std::mutex> lock(dev->mu); *scopes = dev->frag_scopes; }
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_reset_kernel_times(mirhi_device* dev) {
    NULL_CHECK(dev, "device");
    { mirhi_result r0 = sync_all_lanes(dev); if (r0 != MIRHI_OK) return r0; }
    std::lock_guard<std::mutex> lock(dev->mu);
    mirhi_result r = drain_events(dev);
    if (r != MIRHI_OK) return r;
    for (int k = 0; k < MIRHI_KERNEL_COUNT; k++) { dev->total_ms[k] = 0; dev->launches[k] = 0; }
    dev->timeline.clear();
    if (dev->base_stop) { dev->free_events.push_back(dev->base_stop); dev->base_stop = nullptr; }
    dev->base_end_ms = 0.0;
    dev->frag_scopes = 0;
    HIP_TRY(hipMemsetAsync(dev->frag_stats, 0, 2 * sizeof(unsigned long long), dev->stream));
    HIP_TRY(hipStreamSynchronize(dev->stream));
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_get_stats(mirhi_device* dev, mirhi_device_stats* out) {
    NULL_CHECK(dev, "device"); NULL_CHECK(out, "out");
    drain_submits(dev);
    std::lock_guard<std::mutex> lock(dev->mu);
    *out = dev->stats;
    out->native_dispatches = dev->native ? (uint32_t)dev->native->dispatches.load(std::memory_order_relaxed) : 0u;
    out->dispatch_path = 0u;
    if (dev->native && dev->native->ok) {
        out->dispatch_path = 1u;
        for (const NativeQueue* nq : dev->native_lanes) if (nq && nq->proxy) out->dispatch_path = 2u;
    }
    out->device_lost = dev->native && dev->native->lost.load(std::memory_order_acquire) ? 1u : 0u;
    out->reserved = 0u;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_dispatch_path(mirhi_device* dev, char* out, uint32_t out_len) {
    NULL_CHECK(dev, "device"); NULL_CHECK(out, "out");
    if (out_len == 0) return MIRHI_OK;
    std::lock_guard<std::mutex> lock(dev->mu);
    if (dev->native && dev->native->ok) {
        bool proxy = false;
        for (const NativeQueue* nq : dev->native_lanes) proxy |= nq && nq->proxy;
        snprintf(out, out_len, "native: AQL packets on the device's own ROCr queues%s%s", proxy ? " (queue intercepted by a tool: every packet keeps its barrier bit)" : "",
                 (!dev->owns_stream && !dev->native_on_external) ? "; lane 0 stays on the caller's HIP stream" : "");
    } else snprintf(out, out_len, "hip: %s", dev->native ? dev->native->why.c_str() : "native dispatcher not opened");
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_set_native_dispatch(mirhi_device* dev, uint32_t enable) {
    NULL_CHECK(dev, "device");
    mirhi_result r = sync_all_lanes(dev);          // (a lane changes the queue it submits to: nothing may be in flight across the switch)
    if (r != MIRHI_OK) return r;
    std::lock_guard<std::mutex> lock(dev->mu);
    dev->native_on_external = enable != 0;
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_device_measure_roundtrip(mirhi_device* dev, uint32_t lane, uint32_t reps, double* out_us) {
    NULL_CHECK(dev, "device"); NULL_CHECK(out_us, "out_us");
    out_us[0] = out_us[1] = 0.0;
    mirhi_result r = sync_all_lanes(dev);
    if (r != MIRHI_OK) return r;
    std::unique_lock<std::mutex> lock(dev->mu);
    NativeQueue* nq = native_lane(dev, lane);
    if (!nq) return fail(MIRHI_ERR_LOADING, "Loading error: native dispatch unavailable: %s", dev->native ? dev->native->why.c_str() : "not opened");
    lock.unlock();
    if (reps == 0) reps = 1;
    hsa_signal_t sig;
    if (hsa_signal_create(0, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) return fail(MIRHI_ERR_DEVICE, "Vulkan error: hsa_signal_create failed");
    for (int what = 0; what < 2; what++) {
        double total = 0.0;
        for (uint32_t i = 0; i < reps + 8u; i++) {
            hsa_signal_store_relaxed(sig, 1);
            const auto t0 = std::chrono::steady_clock::now();
            if (what == 0) {
                const hipError_t e = launch_noop(nq, sig.handle);
                if (e != hipSuccess) { (void)hsa_signal_destroy(sig); return dev->native->lost.load() ? device_lost(dev) : hip_fail(e, "noop dispatch"); }
            } else {
                std::lock_guard<std::mutex> qlock(nq->mu);
                if (!native_reserve(nq)) { (void)hsa_signal_destroy(sig); return device_lost(dev); }
                auto* p = reinterpret_cast<hsa_barrier_and_packet_t*>(nq->q->base_address) + (nq->widx & (NATIVE_QUEUE_PACKETS - 1));
                memset(reinterpret_cast<uint8_t*>(p) + 4, 0, sizeof *p - 4);
                p->completion_signal = sig;
                native_publish(nq, p, (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                      (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE), 0);
            }
            if (native_signal_wait(sig, UINT64_MAX, dev->native, nq, "round-trip measurement") != 0) { (void)hsa_signal_destroy(sig); return device_lost(dev); }
            if (i >= 8u) total += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        }
        out_us[what] = total / reps;
    }
    (void)hsa_signal_destroy(sig);
    return MIRHI_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-GPU: exchange of the finished bands over RCCL
// ------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
std::mutex g_rccl_mu;
Rccl g_rccl;

// The library a process already holds (PyTorch ships its own librccl.so.1) is found by its soname; otherwise the ROCm install.
mirhi_result rccl_load(Rccl** out) {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    Rccl& R = g_rccl;
    if (!R.handle) {
        const char* env = getenv("MIRHI_RCCL_LIBRARY");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            R.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (R.handle) break;
            R.error = dlerror();
        }
        if (!R.handle) return fail(MIRHI_ERR_LOADING, "Loading error: librccl not found (%s); the tile-row split needs RCCL", R.error.c_str());
        bool ok = true;
#define MIRHI_RCCL_SYM(field, name) do { R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.handle, name)); if (!R.field) { ok = false; R.error = name; } } while (0)
        MIRHI_RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); MIRHI_RCCL_SYM(CommInitRank, "ncclCommInitRank"); MIRHI_RCCL_SYM(CommDestroy, "ncclCommDestroy");
        MIRHI_RCCL_SYM(CommCount, "ncclCommCount"); MIRHI_RCCL_SYM(GroupStart, "ncclGroupStart"); MIRHI_RCCL_SYM(GroupEnd, "ncclGroupEnd");
        MIRHI_RCCL_SYM(Send, "ncclSend"); MIRHI_RCCL_SYM(Recv, "ncclRecv"); MIRHI_RCCL_SYM(Broadcast, "ncclBroadcast");
        MIRHI_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef MIRHI_RCCL_SYM
        if (!ok) { dlclose(R.handle); R.handle = nullptr; return fail(MIRHI_ERR_LOADING, "Loading error: librccl lacks %s", R.error.c_str()); }
    }
    *out = &R;
    return MIRHI_OK;
}
}  // namespace

struct mirhi_comm {
    mirhi_device* dev;
    Rccl* rccl;
    ncclComm_t comm;
    uint32_t rank, world, counted;
    // Every exchange runs on ONE stream of the communicator's own, whatever queue lane rendered the frame: RCCL sees a single,
    // totally ordered stream, and frames in flight on other lanes keep rendering while a band exchange is under way.
    hipStream_t stream;
    hipEvent_t ready, done;       // lane -> exchange stream, exchange stream -> lane
    // Interleaved tile rows: a rank's share of the frame is one 32-row piece per tile row it owns, a tile row of the frame apart.  Sending the pieces one by
    // one would put (world - 1) x 2 x ~9 point-to-point operations into the group for a 4K frame on 8 ranks; instead the share is packed into slot `rank` of
    // this buffer (one strided device-to-device copy), ONE message per peer goes out of it and ONE comes into every other slot -- the op count of the
    // band layout -- and the received slots are scattered into the frame (one strided copy per peer).  [world][slot_bytes], grown on demand.
    uint8_t* pack = nullptr; size_t pack_bytes = 0;
    // A frame that left the library as AQL packets (native dispatch) is followed on its queue by one single-lane kernel that stores the exchange's sequence
    // number into `seq_word` (signal memory), and the exchange stream waits for that value (hipStreamWaitValue64): the stream is ordered behind the
    // frame's raster kernel without the host waiting for anything.
    uint64_t* seq_word = nullptr; uint64_t seq = 0;
};
#define RCCL_TRY(R, expr)                                                                                     \
    do {                                                                                                      \
        ncclResult_t r__ = (expr);                                                                            \
        if (r__ != ncclSuccess) return fail(MIRHI_ERR_DEVICE, "Vulkan error: RCCL: %s (%s)", (R)->GetErrorString(r__), #expr); \
    } while (0)

extern "C" mirhi_result mirhi_comm_unique_id(uint8_t* id) {
    NULL_CHECK(id, "id");
    static_assert(MIRHI_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    Rccl* R = nullptr;
    mirhi_result r = rccl_load(&R);
    if (r != MIRHI_OK) return r;
    ncclUniqueId u;
    RCCL_TRY(R, R->GetUniqueId(&u));
    memcpy(id, u.internal, MIRHI_COMM_ID_BYTES);
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_comm_create(mirhi_device* dev, const uint8_t* id, uint32_t rank, uint32_t world, mirhi_comm** out) {
    NULL_CHECK(dev, "device"); NULL_CHECK(id, "id"); NULL_CHECK(out, "out");
    *out = nullptr;
    if (world == 0 || rank >= world) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: communicator rank %u of %u", rank, world);
    Rccl* R = nullptr;
    mirhi_result r = rccl_load(&R);
    if (r != MIRHI_OK) return r;
    HIP_TRY(hipSetDevice(dev->ordinal));
    ncclUniqueId u;
    memcpy(u.internal, id, MIRHI_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    RCCL_TRY(R, R->CommInitRank(&comm, (int)world, u, (int)rank));
    int counted = 0;
    RCCL_TRY(R, R->CommCount(comm, &counted));
    mirhi_comm* c = new (std::nothrow) mirhi_comm{dev, R, comm, rank, world, (uint32_t)counted, nullptr, nullptr, nullptr};
    if (!c) { (void)R->CommDestroy(comm); return fail(MIRHI_ERR_ALLOCATOR, "Allocator error: host allocation failed"); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError(); (void)R->CommDestroy(comm); delete c;
        return fail(MIRHI_ERR_DEVICE, "Vulkan error: stream / event creation for the band exchange failed");
    }
    // signal memory for ordering the exchange stream behind natively dispatched frames (without it the host waits for the frame instead)
    { void* w = nullptr; if (hipExtMallocWithFlags(&w, 8, hipMallocSignalMemory) == hipSuccess) { c->seq_word = (uint64_t*)w; *c->seq_word = 0; } else (void)hipGetLastError(); }
    { std::lock_guard<std::mutex> lk(dev->mu); dev->aux_streams.push_back(c->stream); }
    dev->split_rank = rank; dev->split_world = world;
    dev->children++;
    *out = c;
    return MIRHI_OK;
}
extern "C" uint32_t mirhi_comm_world(const mirhi_comm* comm) { return comm ? comm->counted : 0; }
extern "C" uint32_t mirhi_comm_rank(const mirhi_comm* comm) { return comm ? comm->rank : 0; }

extern "C" mirhi_result mirhi_comm_all_gather_bands(mirhi_comm* comm, mirhi_image* frame, mirhi_cmd* after, mirhi_gather_algo algo) {
    NULL_CHECK(comm, "comm"); NULL_CHECK(frame, "frame");
    mirhi_device* dev = comm->dev;
    if (frame->dev != dev || (after && after->dev != dev)) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: frame / command buffer belongs to another device");
    if (algo != MIRHI_GATHER_DIRECT && algo != MIRHI_GATHER_BROADCAST) return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: unknown gather algorithm %d", (int)algo);
    if (dev->split_world != comm->world || dev->split_rank != comm->rank)
        return fail(MIRHI_ERR_INVALID_HANDLE, "Invalid handle: the device's tile split (%u of %u) is not the communicator's (%u of %u)", dev->split_rank, dev->split_world, comm->rank, comm->world);
    drain_submits(dev);
    HIP_TRY(hipSetDevice(dev->ordinal));
    hipStream_t lane = after && after->last_stream ? after->last_stream : dev->lanes[after && after->lane < dev->lanes.size() ? after->lane : 0];
    NativeQueue* const native_after = (after && after->last_native && comm->seq_word) ? after->last_native : nullptr;
    if (after && after->last_native && !native_after && !drain_native(dev, after->last_native)) return device_lost(dev);     // (no signal memory: the host waits for the frame)
    hipStream_t stream = comm->stream;
    dev->foreign_writes++;                                   // (peers write the other ranks' rows into this frame)
    const uint32_t tiles_y = (frame->height + TILE - 1) / TILE;
    const size_t row_bytes = (size_t)frame->width * format_bpp(frame->format);
    const size_t tile_row_bytes = (size_t)TILE * row_bytes;
    Rccl* R = comm->rccl;
    // rank r's share: tile rows first + k * step, k < count; the frame's last tile row may be short
    struct Share { uint32_t first, step, count; size_t bytes; };
    auto share = [&](uint32_t r) {
        Share sh{0, 1, 0, 0};
        split_rows(dev->split_layout, r, comm->world, tiles_y, &sh.first, &sh.step, &sh.count);
        for (uint32_t k = 0; k < sh.count; k++) {
            const size_t p0 = (size_t)(sh.first + k * sh.step) * TILE, p1 = std::min<size_t>(p0 + TILE, frame->height);
            sh.bytes += (p1 > p0 ? p1 - p0 : 0) * row_bytes;
        }
        return sh;
    };
    const Share mine = share(comm->rank);
    if (comm->world == 1) return MIRHI_OK;
    const bool packed = dev->split_layout == MIRHI_SPLIT_INTERLEAVED;
    const size_t slot_bytes = ((tiles_y + comm->world - 1) / comm->world) * tile_row_bytes;
    if (packed && comm->pack_bytes < slot_bytes * comm->world) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (comm->pack) { (void)hipFree(comm->pack); comm->pack = nullptr; comm->pack_bytes = 0; }
        HIP_TRY(hipMalloc((void**)&comm->pack, slot_bytes * comm->world));
        comm->pack_bytes = slot_bytes * comm->world;
    }
    // strided copy between a rank's rows in the frame and its slot of the pack buffer (to_slot: gather, else scatter); the last tile row of the frame may be short
    auto strided = [&](const Share& sh, uint8_t* slot, bool to_slot) -> hipError_t {
        if (!sh.count) return hipSuccess;
        uint8_t* rows = frame->ptr + (size_t)sh.first * tile_row_bytes;
        const size_t pitch = (size_t)sh.step * tile_row_bytes;
        const size_t last_p0 = (size_t)(sh.first + (sh.count - 1) * sh.step) * TILE;
        const size_t last_bytes = (std::min<size_t>(last_p0 + TILE, frame->height) - last_p0) * row_bytes;
        const uint32_t full = last_bytes == tile_row_bytes ? sh.count : sh.count - 1;
        hipError_t e = hipSuccess;
        if (full) e = to_slot ? hipMemcpy2DAsync(slot, tile_row_bytes, rows, pitch, tile_row_bytes, full, hipMemcpyDeviceToDevice, stream)
                              : hipMemcpy2DAsync(rows, pitch, slot, tile_row_bytes, tile_row_bytes, full, hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess && full < sh.count && last_bytes)
            e = to_slot ? hipMemcpyAsync(slot + (size_t)full * tile_row_bytes, rows + (size_t)full * pitch, last_bytes, hipMemcpyDeviceToDevice, stream)
                        : hipMemcpyAsync(rows + (size_t)full * pitch, slot + (size_t)full * tile_row_bytes, last_bytes, hipMemcpyDeviceToDevice, stream);
        return e;
    };
    // the exchange starts behind the frame's raster kernel ...
    if (native_after) {
        const uint64_t want = ++comm->seq;
        { const hipError_t ne = launch_seq_store(native_after, comm->seq_word, want); if (ne != hipSuccess) return dev->native->lost.load() ? device_lost(dev) : hip_fail(ne, "sequence store behind the frame"); }
        HIP_TRY(hipStreamWaitValue64(stream, comm->seq_word, want, hipStreamWaitValueGte, ~0ull));
    } else {
        HIP_TRY(hipEventRecord(comm->ready, lane));
        HIP_TRY(hipStreamWaitEvent(stream, comm->ready, 0));
    }
    if (packed) HIP_TRY(strided(mine, comm->pack + comm->rank * slot_bytes, true));
    RCCL_TRY(R, R->GroupStart());
    // Inside the group nothing returns early: a failing call must not leave the thread's RCCL group open (every later RCCL call of
    // this thread, torch.distributed's included, would queue into a group that never ends).  The first error is kept, the group is
    // closed, and only a successful exchange makes the lane wait for it.
    ncclResult_t first = ncclSuccess; const char* what = "";
    auto note = [&](ncclResult_t rc, const char* call) { if (rc != ncclSuccess && first == ncclSuccess) { first = rc; what = call; } };
    uint8_t* const my_ptr = packed ? comm->pack + comm->rank * slot_bytes : frame->ptr + (size_t)mine.first * tile_row_bytes;
    for (uint32_t r = 0; r < comm->world && first == ncclSuccess; r++) {
        const Share sh = share(r);
        uint8_t* const ptr = packed ? comm->pack + r * slot_bytes : frame->ptr + (size_t)sh.first * tile_row_bytes;
        if (algo == MIRHI_GATHER_BROADCAST) {
            if (sh.bytes) note(R->Broadcast(ptr, ptr, sh.bytes, ncclUint8, (int)r, comm->comm, stream), "ncclBroadcast");
            continue;
        }
        if (r == comm->rank) continue;
        if (mine.bytes) note(R->Send(my_ptr, mine.bytes, ncclUint8, (int)r, comm->comm, stream), "ncclSend");
        if (sh.bytes && first == ncclSuccess) note(R->Recv(ptr, sh.bytes, ncclUint8, (int)r, comm->comm, stream), "ncclRecv");
    }
    note(R->GroupEnd(), "ncclGroupEnd");
    if (first != ncclSuccess) return fail(MIRHI_ERR_DEVICE, "Vulkan error: RCCL: %s (%s)", R->GetErrorString(first), what);
    if (packed) for (uint32_t r = 0; r < comm->world; r++) if (r != comm->rank) HIP_TRY(strided(share(r), comm->pack + r * slot_bytes, false));
    // ... and whatever touches the frame next waits for the exchange: a HIP lane through an event, an AQL queue through the frame's attachment record
    // (order_attachments: the exchange stream is this image's last user)
    HIP_TRY(hipEventRecord(comm->done, stream));
    if (!native_after) HIP_TRY(hipStreamWaitEvent(lane, comm->done, 0));
    { std::lock_guard<std::mutex> lk(dev->mu); frame->last_stream = stream; frame->last_native = nullptr; frame->last_cmd = nullptr; frame->last_seq = 0; }
    return MIRHI_OK;
}
extern "C" mirhi_result mirhi_comm_destroy(mirhi_comm* comm) {
    NULL_CHECK(comm, "comm");
    (void)hipSetDevice(comm->dev->ordinal);
    (void)sync_all_lanes(comm->dev);
    (void)hipStreamSynchronize(comm->stream);
    (void)comm->rccl->CommDestroy(comm->comm);
    (void)hipEventDestroy(comm->ready); (void)hipEventDestroy(comm->done);
    { std::lock_guard<std::mutex> lk(comm->dev->mu); auto& v = comm->dev->aux_streams; v.erase(std::remove(v.begin(), v.end(), comm->stream), v.end());
      for (mirhi_image* img : comm->dev->images) if (img->last_stream == comm->stream) img->last_stream = nullptr; }
    (void)hipStreamDestroy(comm->stream);
    if (comm->pack) (void)hipFree(comm->pack);
    if (comm->seq_word) (void)hipFree(comm->seq_word);
    comm->dev->children--;
    delete comm;
    return MIRHI_OK;
}
