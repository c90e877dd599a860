// ABI bookkeeping for libfdn_hip.so.
#include "common.hpp"

#include <atomic>
extern "C" int fdn_abi_version(void) { return 15; }

// Diagnostic switch: 1 = every matrix product of the path on the fp32 MFMA (the round-2 kernels) instead of the split-bf16 forms on
// v_mfma_f32_32x32x16_bf16, so that the cross-stream finding can be bisected.  Process-wide, default 0.
// 2 = the default of ABI 10: the level-2 FDSA tail (fdn_fdsa_out, E in 39..76) on the fp32 MFMA; since round 5 the default (0) runs it on the bf16 pipe.
static std::atomic<int> g_matrix_pipe_mode{0};
extern "C" int fdn_set_matrix_pipe(int mode) {
    if (mode < 0 || mode > 2) return FDN_ERR_ARG;
    g_matrix_pipe_mode.store(mode);
    return FDN_OK;
}
bool fdn_matrix_pipe_f32() { return g_matrix_pipe_mode.load() == 1; }
bool fdn_matrix_pipe_wide() { return g_matrix_pipe_mode.load() == 0; }

// Every host launcher of a kernel that issues v_mfma_f32_32x32x16_bf16 / 32x32x8_bf16 reports here, so that the promise of
// fdn_set_matrix_pipe(1) ("no bf16-MFMA kernel is launched") is a number a test can read.
static std::atomic<long long> g_bf16_launches{0};
void fdn_note_bf16_launch() { g_bf16_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" long fdn_bf16_mfma_launches(void) { return g_bf16_launches.load(); }

extern "C" const char* fdn_error_string(int code) {
    switch (code) {
        case FDN_OK: return "ok";
        case FDN_ERR_ARG: return "invalid argument (null pointer, bad shape or unsupported size)";
        case FDN_ERR_LAUNCH: return "HIP kernel launch failed";
        case FDN_ERR_WORKSPACE: return "workspace too small";
        case FDN_ERR_UNSUPPORTED: return "configuration not supported by this build";
        default: return "unknown error";
    }
}

// ------------------------------------------------------------------------------------------------
// per-device launch state (declared in common.hpp)
// ------------------------------------------------------------------------------------------------
#include <mutex>
#include <map>
#include <utility>

namespace {
std::mutex g_mu;
int g_cus_by_dev[64];                                             // 0 = not queried yet
std::map<std::pair<const void*, int>, size_t> g_lds_limit;         // (kernel, device) -> dynamic LDS bytes already allowed
struct OccKey {
    const void* k; int dev, threads; size_t lds;
    bool operator<(const OccKey& o) const {
        if (k != o.k) return k < o.k;
        if (dev != o.dev) return dev < o.dev;
        if (threads != o.threads) return threads < o.threads;
        return lds < o.lds;
    }
};
std::map<OccKey, int> g_occ;
}  // namespace

bool fdn_occupancy(int* blocks_per_cu, const void* kernel, int threads, size_t lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(g_mu);
    const OccKey key{kernel, dev, threads, lds};
    auto it = g_occ.find(key);
    if (it == g_occ.end()) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds) != hipSuccess) return false;
        it = g_occ.emplace(key, n).first;
    }
    *blocks_per_cu = it->second;
    return true;
}

int fdn_device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_cus_by_dev[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return -1;
        g_cus_by_dev[dev] = n;
    }
    return g_cus_by_dev[dev];
}

bool fdn_allow_dynamic_lds(const void* kernel, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(g_mu);
    size_t& have = g_lds_limit[std::make_pair(kernel, dev)];
    if (have >= bytes) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    have = bytes;
    return true;
}
