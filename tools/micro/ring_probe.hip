// (round 6) Does a tile-local hand-off that is REWRITTEN in place stay in the 256 MB Infinity Cache?  512 resident workgroups (two per CU) each write a
// 152 KB block, read it back (sc1: L2 / beyond) and repeat; either every round goes to a fresh block of a 4.6 GB tensor (the layout of
// fdn_fdsa_fused_tail as first built) or every workgroup rewrites its own block of an 80 MB ring.  Same instructions, same bytes through L2;
// the difference is what reaches HBM.  Also: read-back only from a region written long ago (cold) for the latency / rate contrast.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ring_probe.hip -o gpurun_out/ring_probe && gpurun_out/ring_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
// BLK: bytes per tile block - 152 KB = 4E planes x 1 KB at level 1 (E = 38, two workgroups per CU); 612 KB = the 612 planes of a level-3 tile
// (E = 153, VERDICT r5 item 5: one workgroup per CU walking all 20 channel groups of an 8 x 32 tile and feeding the LN3 GEMM from its own block)
template <bool RING, bool READ, int BLK, int WGS>
__global__ __launch_bounds__(256, WGS) void k(char* buf, int rounds, unsigned* sink) {
    const int tid = threadIdx.x;
    unsigned acc = 0;
    for (int r = 0; r < rounds; ++r) {
        const long tile = RING ? blockIdx.x : (long)r * gridDim.x + blockIdx.x;
        char* base = buf + tile * BLK;
        const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, BLK, 0x00020000);
        for (int i = 0; i < BLK / 4096; ++i)            // 38 x 4 KB: a wave writes 1 KB rows
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{(unsigned)r, (unsigned)i, (unsigned)tid, acc}, rs, i * 4096 + tid * 16, 0, 0);
        if (READ) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int i = 0; i < BLK / 4096; ++i) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 4096 + ((tid + 64) & 255) * 16, 0, 16);
                acc += v.x ^ v.w;
            }
            __syncthreads();
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int BLK, int WGS>
void probe(int rounds) {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int grid = pr.multiProcessorCount * WGS;
    char* buf; unsigned* sink;
    const size_t big = (size_t)grid * rounds * BLK;
    hipMalloc(&buf, big); hipMalloc(&sink, 64);
    hipMemset(buf, 0, big);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern) {
        float best = 1e9f, ms = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, rounds, sink);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double gb = (double)grid * rounds * BLK / 1e9;
        printf("%-44s %.3f ms   %.2f GB written%s  -> %.2f TB/s per direction\n", name, best, gb, "", gb / best);
    };
    printf("grid %d workgroups (%d per CU) x %d rounds x %d KB: ring %.0f MB, fresh %.2f GB\n", grid, WGS, rounds, BLK / 1024, grid * (double)BLK / 1e6, big / 1e9);
    run("fresh blocks, write only", k<false, false, BLK, WGS>);
    run("ring (rewritten in place), write only", k<true, false, BLK, WGS>);
    run("fresh blocks, write + same-workgroup read-back", k<false, true, BLK, WGS>);
    run("ring, write + same-workgroup read-back", k<true, true, BLK, WGS>);
    hipFree(buf); hipFree(sink);
}
int main() {
    probe<152 * 1024, 2>(57);          // level 1: 80 MB in flight
    probe<612 * 1024, 1>(14);          // level 3, one workgroup per CU: 160 MB in flight
    probe<612 * 1024, 2>(7);           // level 3, two per CU: 321 MB in flight (beyond the Infinity Cache)
    return 0;
}
