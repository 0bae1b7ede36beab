// Device functions shared by the level kernels (pqd_kernels.hip, pqd_levels.hip): the hybrid-RLE run header
// step, guarded loads, the bit-unpackers and the 16-value run decode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pqd_device.h"

namespace pqd {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
// A pointer that was read from a table in memory is a generic one to the compiler, and stores through it become FLAT
// instructions (checked against the LDS aperture, counted on the LDS counter as well, so that LDS reads behind them
// wait for them).  Output arrays always lie in device memory: as_global says so.
#define PQD_GLOBAL __attribute__((address_space(1)))
template <class T>
static __device__ __forceinline__ PQD_GLOBAL T* as_global(T* p) { return (PQD_GLOBAL T*)p; }
typedef u64 u64_u __attribute__((aligned(1)));
typedef u32 u32_u __attribute__((aligned(1)));

static __device__ __forceinline__ void set_status(int* status, int code, u32 where) {
    if (atomicCAS(status, 0, code) == 0) status[1] = (int)where;
}

static __device__ __forceinline__ u32 sat_add32(u32 a, u32 b) {
    const u32 s = a + b;
    return s < a ? 0xFFFFFFFFu : s;
}

// One hybrid-RLE run header at region position p (p < L), given the 8 bytes at p..p+7 as two little-endian
// words (bytes at or past L read as zero).  nxt = position of the next header (== L when the block ends),
// ERR_POS when the reference would fail:
//   header = ULEB128 (readLEB128, rle.go:274-290; headers longer than 8 bytes are rejected here)
//   even → RLE run: count = header >> 1, value in ceil(w/8) bytes (rle.go:218-244)
//   odd  → bit-packed: groups = header >> 1, values = 8*groups, w*groups bytes; a short read is zero
//          padded, a read at the very end of the block is io.EOF (rle.go:197-216)
// info = (position of the value byte(s) or of the packed bytes) << 1 | bit-packed flag.
// Runs of more than 2^31 values and bit-packed runs of more than 2^28 groups are rejected, so only the
// low 32 header bits are ever assembled; the bits above them just have to be zero (or, for an RLE run
// of exactly 2^31 values, bit 32 alone).  Everything is 32-bit arithmetic: this is the inner loop of
// the chain kernels.
static __device__ __forceinline__ void rle_step_w32(u32 lo, u32 hi, u32 p, u32 L, u32 w, u32& nxt, u32& vals, u32& info) {
    // straight-line on purpose: selects instead of early returns keep the wave out of exec-mask branches
    const u32 slo = ~lo & 0x80808080u, shi = ~hi & 0x80808080u;
    const bool in_lo = slo != 0;
    const u32 s = in_lo ? slo : shi;                      // stop bit lives in this word
    const u32 n = ((u32)__builtin_ctz(s | 0x80000000u) >> 3) + (in_lo ? 1u : 5u);  // header bytes
    const u32 m = s ^ (s - 1);                            // every bit up to the stop byte
    const u32 a = lo & (in_lo ? m : 0xFFFFFFFFu), b = hi & (in_lo ? 0u : m) & 0x7F7F7F7Fu;
    bool bad = (s == 0) | (n > L - p);                    // no stop byte in 8 / header cut off by the block end
    const u32 h = (a & 0x7F) | (((a >> 8) & 0x7F) << 7) | (((a >> 16) & 0x7F) << 14) | (((a >> 24) & 0x7F) << 21) | (b << 28);
    const u32 above = b & 0x7F7F7F70u;                    // header bits 32 and up
    const u32 q = p + n;
    const bool bp = (h & 1) != 0;
    const u32 cnt = h >> 1;
    // RLE run
    const u32 bw = (w + 7) >> 3;
    const bool rle_bad = ((above != 0) & ((above != 0x10u) | (cnt != 0))) | (bw > L - q);  // > 2^31 values / io.EOF on the value
    const u32 rle_cnt = above ? 0x80000000u : cnt;
    // bit-packed run
    const bool bp_bad = (above != 0) | (cnt > (1u << 28)) | ((w != 0) & (q >= L)); // exhausted reader: io.EOF
    u32 bp_nxt;
    if (L < (1u << 24)) {
        // w * groups in 24-bit arithmetic: groups beyond 2^24 - 1 overshoot the block anyway
        const u32 prod = __umul24(w, cnt < 0xFFFFFFu ? cnt : 0xFFFFFFu);
        bp_nxt = prod >= L - q ? L : q + prod;
    } else {
        const u64 e = (u64)q + (u64)w * cnt;
        bp_nxt = e > L ? L : (u32)e;
    }
    if (w == 0) bp_nxt = q;
    bad = bad | (bp ? bp_bad : rle_bad);
    nxt = bad ? ERR_POS : (bp ? bp_nxt : q + bw);
    vals = bad ? 0u : (bp ? cnt * 8 : rle_cnt);
    info = bad ? 0u : ((q << 1) | (bp ? 1u : 0u));
}
static __device__ __forceinline__ void rle_step_win(u64 x, u32 p, u32 L, u32 w, u32& nxt, u64& vals, u32& info) {
    u32 v;
    rle_step_w32((u32)x, (u32)(x >> 32), p, L, w, nxt, v, info);
    vals = v;
}

static __device__ __forceinline__ u64 load_u64_guarded(const u8* p, u32 pos, u32 L) {
    if (pos < L && L - pos >= 8) return *(const u64_u*)(p + pos);
    u64 v = 0;
    for (u32 k = 0; k < 8; k++) if (pos + k < L) v |= (u64)p[pos + k] << (8 * k);
    return v;
}

static __device__ __forceinline__ u32x4 load_u128_guarded(const u8* p, u32 pos, u32 L) {
    if (pos < L && L - pos >= 16) return *(const u32x4_u*)(p + pos);
    u32 t[4] = {0, 0, 0, 0};
    for (u32 k = 0; k < 16; k++) if (pos + k < L) t[k >> 2] |= (u32)p[pos + k] << (8 * (k & 3));
    return u32x4{t[0], t[1], t[2], t[3]};
}

// 16 consecutive values (one per result byte) of a bit-packed run of W-bit levels, starting at value rel0
// of the run whose packed bytes begin at region position pay (bitpack.Unpack, bitpack.go:76-141: value j of
// a group of 8 sits at bit j*W of the group's W bytes, LSB first; a short read is zero padded, rle.go:203-213).
// 16 values take at most 7 + 16*W <= 128 bits from their first byte, so one 16-byte load covers them.
template <int W>
static __device__ __forceinline__ u32x4 unpack16(const u8* rsrc, u32 L, u32 pay, u32 rel0) {
    const u32 bit = (rel0 & 7) * W;
    u32x4 x = load_u128_guarded(rsrc, pay + (rel0 >> 3) * W + (bit >> 3), L);
    if (W == 8) return x;
    const u32 sh = bit & 7;
    u32 d[5];
    d[0] = __builtin_amdgcn_alignbit(x.y, x.x, sh);
    d[1] = __builtin_amdgcn_alignbit(x.z, x.y, sh);
    d[2] = __builtin_amdgcn_alignbit(x.w, x.z, sh);
    d[3] = x.w >> sh;
    d[4] = 0;
    u32 o[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int b = k * W, q = b >> 5, r = b & 31;
        u32 v;
        if (r + W <= 32) v = (d[q] >> r) & ((1u << W) - 1);
        else v = __builtin_amdgcn_alignbit(d[q + 1], d[q], r) & ((1u << W) - 1);
        o[k >> 2] |= v << (8 * (k & 3));
    }
    return u32x4{o[0], o[1], o[2], o[3]};
}
// 16 values from position `from` on of the run (out_off, info), width w <= 8
static __device__ __forceinline__ u32x4 run16(const u8* rsrc, u32 L, u32 w, u32 out_off, u32 info, u32 from) {
    if (!(info & 1)) {
        const u32 v = (info >> 1) * 0x01010101u;  // RLE: the record carries the value
        return u32x4{v, v, v, v};
    }
    const u32 pay = info >> 1, rel0 = from - out_off;
    switch (w) {
    case 1: return unpack16<1>(rsrc, L, pay, rel0);
    case 2: return unpack16<2>(rsrc, L, pay, rel0);
    case 3: return unpack16<3>(rsrc, L, pay, rel0);
    case 4: return unpack16<4>(rsrc, L, pay, rel0);
    case 5: return unpack16<5>(rsrc, L, pay, rel0);
    case 6: return unpack16<6>(rsrc, L, pay, rel0);
    case 7: return unpack16<7>(rsrc, L, pay, rel0);
    default: return unpack16<8>(rsrc, L, pay, rel0);
    }
}

// Bytes k0.. of the 16-byte accumulator (lo, hi) are replaced by the leading bytes of v.
static __device__ __forceinline__ void merge16(u64& lo, u64& hi, const u32x4 v, u32 k0) {
    const u64 vlo = v.x | ((u64)v.y << 32), vhi = v.z | ((u64)v.w << 32);
    if (k0 == 0) { lo = vlo; hi = vhi; }
    else if (k0 < 8) {
        const u32 sh = 8 * k0;
        lo = (lo & ((1ull << sh) - 1)) | (vlo << sh);
        hi = (vhi << sh) | (vlo >> (64 - sh));
    } else {
        const u32 sh = 8 * (k0 - 8);
        hi = (hi & ((1ull << sh) - 1)) | (vlo << sh);
    }
}

}  // namespace pqd
