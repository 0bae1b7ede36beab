// Micro-benchmark for the PLAIN value gather: variants of the segment-copy kernel, interleaved rounds in
// one process on the same buffers (misaligned page payloads as in a real file).
//   hipcc --offload-arch=gfx950 -O3 -o tools/copy_bench tools/copy_bench.hip && ./tools/copy_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
struct Seg { u64 src_off; u8* dst; u64 len; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ u32 find_owner(const u32* p, u32 n, u32 x) {
    u32 lo = 0, hi = n;
    while (hi - lo > 1) { u32 mid = (lo + hi) >> 1; if (p[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}

// MODE 0: unaligned dwordx4 loads; MODE 1: dword-aligned loads + v_alignbyte with neighbour shuffle
template <int UNROLL, bool NT_LD, bool NT_ST, int MODE, int THREADS = 256, bool TABLE = false>
__global__ __launch_bounds__(THREADS) void k_copy(const u8* __restrict__ src_base, const Seg* __restrict__ segs,
                                              const u32* __restrict__ blk_prefix, u32 n_segs, u32 n_blocks) {
    constexpr u32 BLK = THREADS * 16 * UNROLL;
    const u32 tid = threadIdx.x;
    for (u32 b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        // TABLE: blk_prefix[n_segs + 1 + b] holds the segment of block b
        const u32 s = TABLE ? blk_prefix[n_segs + 1 + b] : find_owner(blk_prefix, n_segs, b);
        const Seg sg = segs[s];
        const u32 bi = b - blk_prefix[s];
        const u64 a = (u64)sg.dst & 15;
        const u8* sv = src_base + sg.src_off - a;
        u8* dv = sg.dst - a;
        const u64 v_end = a + sg.len;
        u64 v_lo = (u64)bi * BLK;
        u64 v_hi = v_lo + BLK < v_end ? v_lo + BLK : v_end;
        if (v_lo < a) v_lo = a;
        const u64 u0 = (v_lo + 15) & ~15ull, u1 = v_hi & ~15ull;
        if (u1 > u0) {
            const u32 n_units = (u32)((u1 - u0) >> 4);
            u32x4 r[UNROLL];
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < UNROLL; i++) {
                    const u32 k = tid + i * THREADS;
                    if (k < n_units) {
                        const u32x4_u* p = (const u32x4_u*)(sv + u0 + (u64)k * 16);
                        r[i] = NT_LD ? __builtin_nontemporal_load(p) : *p;
                    }
                }
            } else {
                const u32 sh = (u32)((u64)(sv + u0) & 3);
                const u8* sal = sv + u0 - sh;  // dword aligned
#pragma unroll
                for (int i = 0; i < UNROLL; i++) {
                    const u32 k = tid + i * THREADS;
                    u32x4 v = {0, 0, 0, 0};
                    u32 extra = 0;
                    if (k < n_units) {
                        const u32x4* p = (const u32x4*)(sal + (u64)k * 16);  // 4-byte aligned only
                        v = NT_LD ? __builtin_nontemporal_load((const u32x4_u*)p) : *(const u32x4_u*)p;
                    }
                    u32 nb = __shfl_down(v.x, 1);
                    if ((tid & 63) == 63 && k < n_units && sh) extra = *(const u32*)(sal + (u64)k * 16 + 16);
                    if ((tid & 63) == 63) nb = extra;
                    u32x4 o;
                    o.x = __builtin_amdgcn_alignbyte(v.y, v.x, sh);
                    o.y = __builtin_amdgcn_alignbyte(v.z, v.y, sh);
                    o.z = __builtin_amdgcn_alignbyte(v.w, v.z, sh);
                    o.w = __builtin_amdgcn_alignbyte(nb, v.w, sh);
                    r[i] = o;
                }
            }
#pragma unroll
            for (int i = 0; i < UNROLL; i++) {
                const u32 k = tid + i * THREADS;
                if (k < n_units) {
                    u32x4* q = (u32x4*)(dv + u0 + (u64)k * 16);
                    if (NT_ST) __builtin_nontemporal_store(r[i], q); else *q = r[i];
                }
            }
            if (tid < (u32)(u0 - v_lo)) dv[v_lo + tid] = sv[v_lo + tid];
            if (tid < (u32)(v_hi - u1)) dv[u1 + tid] = sv[u1 + tid];
        } else if (tid < (u32)(v_hi - v_lo)) dv[v_lo + tid] = sv[v_lo + tid];
    }
}

// reference: plain aligned 16-byte grid-stride copy of one contiguous buffer
__global__ __launch_bounds__(256) void k_plain(const u32x4* __restrict__ s, u32x4* __restrict__ d, u64 n) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) d[i] = s[i];
}

struct Variant { const char* name; int unroll; void (*fn)(const u8*, const Seg*, const u32*, u32, u32); u32 grid; int threads; };

int main(int argc, char** argv) {
    const u64 rows = 1000000;
    const int pages = argc > 1 ? atoi(argv[1]) : 32, rgs = 2;
    // file layout: 3 columns (4, 8, 8 bytes), header ~21 bytes before each page
    std::vector<u64> page_off, page_len;
    u64 pos = 4;
    for (int g = 0; g < rgs; g++)
        for (int c = 0; c < 3; c++)
            for (int p = 0; p < pages; p++) {
                pos += 21 + (p % 3);
                u64 len = rows * (c == 0 ? 4 : 8);
                page_off.push_back(pos); page_len.push_back(len);
                pos += len;
            }
    const u64 total = pos;
    u8 *d_src, *d_dst;
    CK(hipMalloc(&d_src, total + 64));
    CK(hipMalloc(&d_dst, total + 4096));
    std::vector<u8> h(total);
    u64 x = 88172645463325252ull;
    for (u64 i = 0; i + 8 <= total; i += 8) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; *(u64*)&h[i] = x; }
    CK(hipMemcpy(d_src, h.data(), total, hipMemcpyHostToDevice));
    std::vector<Seg> segs;
    u64 o = 0;
    for (size_t i = 0; i < page_off.size(); i++) { segs.push_back({page_off[i], d_dst + o, page_len[i]}); o += page_len[i]; }
    const u64 payload = o;
    Seg* d_segs; CK(hipMalloc(&d_segs, segs.size() * sizeof(Seg)));
    CK(hipMemcpy(d_segs, segs.data(), segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
    // key = bytes per block / 1024
    auto prefix_for = [&](u64 BLK) {
        std::vector<u32> p{0};
        for (auto& s : segs) p.push_back(p.back() + (u32)((((u64)s.dst & 15) + s.len + BLK - 1) / BLK));
        const u32 nb = p.back();
        for (u32 sidx = 0; sidx < segs.size(); sidx++)
            for (u32 b = p[sidx]; b < p[sidx + 1]; b++) p.push_back(sidx);
        (void)nb;
        return p;
    };
    u32* d_prefix[65] = {};
    u32 nblocks[65] = {};
    for (int kb : {1, 2, 4, 8, 16, 32, 64}) {
        auto p = prefix_for((u64)kb * 1024);
        nblocks[kb] = p[segs.size()];
        CK(hipMalloc(&d_prefix[kb], p.size() * 4));
        CK(hipMemcpy(d_prefix[kb], p.data(), p.size() * 4, hipMemcpyHostToDevice));
    }
    std::vector<Variant> vs = {
        {"t256 u4 g2048 (current)", 4, k_copy<4, true, true, 0>, 2048, 256},
        {"t256 u4 gALL", 4, k_copy<4, true, true, 0>, 0, 256},
        {"t256 u2 gALL", 2, k_copy<2, true, true, 0>, 0, 256},
        {"t256 u1 gALL", 1, k_copy<1, true, true, 0>, 0, 256},
        {"t256 u2 gALL table", 2, k_copy<2, true, true, 0, 256, true>, 0, 256},
        {"t256 u1 gALL table", 1, k_copy<1, true, true, 0, 256, true>, 0, 256},
        {"t256 u4 gALL table", 4, k_copy<4, true, true, 0, 256, true>, 0, 256},
        {"t128 u2 gALL table", 2, k_copy<2, true, true, 0, 128, true>, 0, 128},
        {"t128 u4 gALL table", 4, k_copy<4, true, true, 0, 128, true>, 0, 128},
        {"t64 u4 gALL table", 4, k_copy<4, true, true, 0, 64, true>, 0, 64},
        {"t64 u8 gALL table", 8, k_copy<8, true, true, 0, 64, true>, 0, 64},
        {"t512 u1 gALL table", 1, k_copy<1, true, true, 0, 512, true>, 0, 512},
        {"t512 u2 gALL table", 2, k_copy<2, true, true, 0, 512, true>, 0, 512},
        {"t1024 u1 gALL table", 1, k_copy<1, true, true, 0, 1024, true>, 0, 1024},
        {"t256 u2 gALL table ld/st", 2, k_copy<2, false, false, 0, 256, true>, 0, 256},
        {"t256 u2 g16384 table", 2, k_copy<2, true, true, 0, 256, true>, 16384, 256},
        {"t256 u2 g65536 table", 2, k_copy<2, true, true, 0, 256, true>, 65536, 256},
    };

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rounds = 12;
    std::vector<std::vector<float>> ms(vs.size() + 2);
    for (int r = 0; r < rounds; r++) {
        for (size_t v = 0; v < vs.size(); v++) {
            const int kb = vs[v].threads * 16 * vs[v].unroll / 1024;
            const u32 nb = nblocks[kb];
            const u32 grid = vs[v].grid ? std::min(vs[v].grid, nb) : nb;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(vs[v].fn, dim3(grid), dim3(vs[v].threads), 0, 0, d_src, d_segs, d_prefix[kb], (u32)segs.size(), nb);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (r >= 2) ms[v].push_back(t);
        }
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_plain, dim3(2048), dim3(256), 0, 0, (const u32x4*)d_src, (u32x4*)d_dst, payload / 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 2) ms[vs.size()].push_back(t);
        CK(hipEventRecord(e0));
        CK(hipMemcpyAsync(d_dst, d_src, payload, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 2) ms[vs.size() + 1].push_back(t);
    }
    // verify the last variant's output
    std::vector<u8> out(payload);
    CK(hipMemset(d_dst, 0, payload));
    hipLaunchKernelGGL(vs[5].fn, dim3(nblocks[4]), dim3(256), 0, 0, d_src, d_segs, d_prefix[4], (u32)segs.size(), nblocks[4]);
    CK(hipMemcpy(out.data(), d_dst, payload, hipMemcpyDeviceToHost));
    u64 bad = 0, oo = 0;
    for (size_t i = 0; i < page_off.size(); i++) { bad += memcmp(&out[oo], &h[page_off[i]], page_len[i]) != 0; oo += page_len[i]; }
    printf("verify table variant: %llu bad pages of %zu\n", (unsigned long long)bad, page_off.size());
    printf("payload %.1f MB, algorithmic bytes per launch %.1f MB\n", payload / 1e6, 2 * payload / 1e6);
    for (size_t v = 0; v < ms.size(); v++) {
        auto m = ms[v]; std::sort(m.begin(), m.end());
        const char* name = v < vs.size() ? vs[v].name : (v == vs.size() ? "REF aligned float4 grid-stride copy" : "REF hipMemcpyDtoD");
        printf("%-36s median %.4f ms  min %.4f ms  -> %.0f GB/s (median)  %.0f GB/s (best)\n", name, m[m.size() / 2], m[0],
               2 * payload / (m[m.size() / 2] * 1e-3) / 1e9, 2 * payload / (m[0] * 1e-3) / 1e9);
    }
    return 0;
}
