// HIP kernels for gfx950 (MI355X): Parquet DATA_PAGE v1 decode — hybrid RLE/bit-packed levels and
// PLAIN values.  Everything here is byte/integer movement bounded by HBM bandwidth; there is no
// contraction anywhere, so no MFMA.  Wavefront = 64 lanes; workgroups are 256 threads.
//
// Reference behaviour each kernel reproduces (paths inside parsyl/parquet):
//   k_copy            binary.Read of PLAIN little-endian values + page concatenation
//                     (cmd/parquetgen/gen/template_required.go:31-33, template_optional.go:48-50,
//                      fields.go:131,313)
//   k_chain<LEVELS>   run-header walk of rle.Read (internal/rle/rle.go:175-193, readLEB128 :274-290)
//   k_levels_expand   readRLE :218-230, readRLEBitPacked :197-216, bitpack.Unpack
//                     (internal/bitpack/bitpack.go:76-141), truncation to NumValues (fields.go:300,308),
//                     valsFromDefs (fields.go:220-228)
//   k_bools           GetBools / unpackBools (parquet.go:455-505)
//   k_chain<STRINGS>, k_str_gather
//                     the [int32 length][bytes] walk of template_string.go:47-58 /
//                     template_string_optional.go:68-79
//   k_rows_*          per-row level consumption of the generated write functions
//                     (cmd/parquetgen/dremel/write_repeated.go:62-80, write_optional.go:22-31)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#include "pqd_device.h"
#include "pqd_rle.h"

namespace pqd {


// Largest i in [0, n) with prefix[i] <= x; prefix has n + 1 non-decreasing entries, prefix[0] == 0.
static __device__ __forceinline__ u32 find_owner(const u32* __restrict__ prefix, u32 n, u32 x) {
    u32 lo = 0, hi = n;
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (prefix[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

// =====================================================================================
// k_copy — gathers byte segments (the values regions of pages) into dense output arrays.
// One block moves CP_BLK bytes of one segment.  Stores are 16-byte aligned dwordx4; loads are
// dwordx4 at whatever alignment the page payload has in the file (gfx950 global loads need no
// alignment).  Each wave instruction covers 1 KiB of contiguous memory on both sides.
// =====================================================================================
#ifndef CP_LOAD_NT
#define CP_LOAD_NT 0
#endif
#ifndef CP_STORE_NT
#define CP_STORE_NT 1
#endif
__global__ __launch_bounds__(CP_THREADS) void k_copy(const u8* __restrict__ src_base, const Seg* __restrict__ segs,
                                                     const u32* __restrict__ blk_prefix, const u32* __restrict__ blk_grp,
                                                     u32 n_segs, u32 n_blocks) {
    const u32 tid = threadIdx.x;
    for (u32 b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        // the segment of block b lies between those of blocks b / CP_GRP * CP_GRP and the next multiple, which a
        // table holds.  Pages are thousands of blocks long, so the two are mostly the same segment and the search
        // ends at once (a binary search over all segments is ten dependent scalar loads, a third of a copy wave's life)
        u32 s = blk_grp[b / CP_GRP], s_hi = blk_grp[b / CP_GRP + 1];
        while (s < s_hi) {
            const u32 mid = (s + s_hi + 1) >> 1;
            if (blk_prefix[mid] <= b) s = mid; else s_hi = mid - 1;
        }
        const Seg sg = segs[s];
        const u32 bi = b - blk_prefix[s];
        const u64 a = (u64)sg.dst & 15;            // destination misalignment
        // v-space: v = offset in segment + a, so that v % 16 == 0 ⇔ destination 16-byte aligned
        const u8* sv = src_base + sg.src_off - a;
        u8* dv = sg.dst - a;
        const u64 v_end = a + sg.len;
        u64 v_lo = (u64)bi * CP_BLK;
        u64 v_hi = v_lo + CP_BLK < v_end ? v_lo + CP_BLK : v_end;
        if (v_lo < a) v_lo = a;
        const u64 u0 = (v_lo + 15) & ~15ull;
        const u64 u1 = v_hi & ~15ull;
        if (u1 > u0) {
            const u32 n_units = (u32)((u1 - u0) >> 4);
            u32x4 r[CP_UNROLL];
#pragma unroll
            for (int i = 0; i < CP_UNROLL; i++) {
                const u32 k = tid + i * CP_THREADS;
#if CP_LOAD_NT
                if (k < n_units) r[i] = __builtin_nontemporal_load((const u32x4_u*)(sv + u0 + (u64)k * 16));
#else
                if (k < n_units) r[i] = *(const u32x4_u*)(sv + u0 + (u64)k * 16);
#endif
            }
#pragma unroll
            for (int i = 0; i < CP_UNROLL; i++) {
                const u32 k = tid + i * CP_THREADS;
#if CP_STORE_NT
                if (k < n_units) __builtin_nontemporal_store(r[i], as_global((u32x4*)(dv + u0 + (u64)k * 16)));
#else
                if (k < n_units) *as_global((u32x4*)(dv + u0 + (u64)k * 16)) = r[i];
#endif
            }
            // head [v_lo, u0) and tail [u1, v_hi): fewer than 16 bytes each
            if (tid < (u32)(u0 - v_lo)) *as_global(dv + v_lo + tid) = sv[v_lo + tid];
            if (tid < (u32)(v_hi - u1)) *as_global(dv + u1 + tid) = sv[u1 + tid];
        } else {
            // fewer than 31 bytes in total
            if (tid < (u32)(v_hi - v_lo)) *as_global(dv + v_lo + tid) = sv[v_lo + tid];
        }
    }
}

// =====================================================================================
// Chain resolution.  Both run headers of a hybrid-RLE block and PLAIN BYTE_ARRAY values form a
// chain: the position of element k+1 is only known after decoding element k.  Three kernels turn
// the chain into parallel work.  A tile is CH_T = 64 x CH_SUB source bytes and belongs to one
// wavefront; lane l owns the CH_SUB-byte sub-tile l.
//   k_chain<.., false>   Treats EVERY byte as a possible element start.  Each lane sweeps its
//                        sub-tile backwards once: the successor of a start either leaves the sub-tile
//                        or lands on a later start of the same sub-tile whose answer is already
//                        memoised, so every start learns in O(1) where its chain leaves the sub-tile
//                        and how many elements / level values it passes.  A second backward sweep over
//                        the 64 sub-tiles turns these into tile exits.  The first E starts of each
//                        tile go to an exit table.
//   k_chain_groups / k_chain_resolve / k_chain_tiles
//                        hop from tile to tile through the exit table (in three levels) and record
//                        where the true chain enters each tile.
//   k_chain<.., true>    same per-lane sweep; lane 0 then follows the true chain from sub-tile to
//                        sub-tile (<= 64 hops), and every lane walks its sub-tile from its true entry,
//                        emitting one record per true element.
// Each lane keeps its sub-tile bytes in registers (the sweeps are fully unrolled, so every header is
// parsed with constant funnel shifts); the memo tables live in LDS at index offset*65 + lane, so lanes
// working on the same offset of different sub-tiles hit different banks.
// =====================================================================================

struct GlobalBytes {
    const u8* p;
    __device__ __forceinline__ u32 operator()(u32 q) const { return p[q]; }
};
static __device__ __forceinline__ u32 memo_idx(u32 rel) { return (rel % CH_SUB) * 65 + rel / CH_SUB; }


// One PLAIN BYTE_ARRAY value at region position p given its 4 length bytes: int32 LE length, then the
// bytes (template_string.go:47-58).  A negative length panics in the reference; a value running past
// the region is reported as corrupt.
static __device__ __forceinline__ void str_step_win(u32 len, u32 p, u32 L, u32& nxt) {
    const u32 rem = L - p;  // p < L
    const bool bad = (rem < 4) | ((len & 0x80000000u) != 0) | (len > rem - 4);
    nxt = bad ? ERR_POS : p + 4 + len;
}

// The same steps reading their bytes one at a time (used off the fast path).
template <class Bytes>
static __device__ __forceinline__ void rle_step(const Bytes& B, u32 p, u32 L, u32 w, u32& nxt, u64& vals, u32& info) {
    u64 x = 0;
    for (u32 k = 0; k < 8 && p + k < L; k++) {
        const u64 b = B(p + k);
        x |= b << (8 * k);
        if (!(b & 0x80)) break;
    }
    rle_step_win(x, p, L, w, nxt, vals, info);
}
template <class Bytes>
static __device__ __forceinline__ void str_step(const Bytes& B, u32 p, u32 L, u32& nxt) {
    u32 len = 0;
    if (L - p >= 4) len = B(p) | (B(p + 1) << 8) | (B(p + 2) << 16) | (B(p + 3) << 24);
    str_step_win(len, p, L, nxt);
}

// One snappy element at region position p (p < L), given the 8 bytes at p (decode of golang/snappy,
// called by pageData, fields.go:361-371):
//   tag & 3 == 0  literal: (tag >> 2) + 1 bytes, or 1..4 extra little-endian length bytes for 60..63
//   tag & 3 == 1  copy: 4 + ((tag >> 2) & 7) bytes from offset ((tag & 0xE0) << 3) | next byte
//   tag & 3 == 2/3 copy: 1 + (tag >> 2) bytes from the offset in the next 2 / 4 bytes
// nxt = position of the next element, vals = bytes it produces,
// info = (position of the literal bytes) << 1, or (offset << 1) | 1 for a copy.  A cut-off element, a
// literal running past the block, offset 0 and offsets of 2^31 and more are ERR_POS; whether a copy
// reaches back before the start of the output is checked where the output position is known.
static __device__ __forceinline__ void snappy_step_w32(u32 lo, u32 hi, u32 p, u32 L, u32& nxt, u32& vals, u32& info) {
    const u32 tag = lo & 0xFF, type = tag & 3, x6 = tag >> 2;
    const u32 rem = L - p;
    const u32 b1234 = (lo >> 8) | (hi << 24);  // the 4 bytes after the tag
    // literal
    const u32 nb = x6 < 60 ? 0 : x6 - 59;
    const u32 ext = nb == 4 ? b1234 : b1234 & ((1u << (8 * nb)) - 1);
    const u32 lit_len = (nb ? ext : x6) + 1;  // 0 after overflow: an unsupported literal length
    // copies
    const u32 c_len = type == 1 ? 4 + (x6 & 7) : 1 + x6;
    const u32 c_off = type == 1 ? ((tag & 0xE0) << 3) | (b1234 & 0xFF) : type == 2 ? b1234 & 0xFFFF : b1234;
    const u32 hdr = type == 0 ? 1 + nb : type == 1 ? 2 : type == 2 ? 3 : 5;
    const bool lit = type == 0;
    bool bad = hdr > rem;
    bad = bad | (lit ? ((lit_len == 0) | (lit_len > rem - hdr)) : ((c_off == 0) | (c_off >= 0x80000000u)));
    nxt = bad ? ERR_POS : (lit ? p + hdr + lit_len : p + hdr);
    vals = bad ? 0u : (lit ? lit_len : c_len);
    info = bad ? 0u : (lit ? (p + hdr) << 1 : (c_off << 1) | 1u);
}

// the element step of a chain kind
template <int KIND>
static __device__ __forceinline__ void elem_step_w32(u32 lo, u32 hi, u32 p, u32 L, u32 w, u32& nxt, u32& vals, u32& info) {
    if (KIND == KIND_SNAPPY) snappy_step_w32(lo, hi, p, L, nxt, vals, info);
    else rle_step_w32(lo, hi, p, L, w, nxt, vals, info);
}
template <int KIND, class Bytes>
static __device__ __forceinline__ void elem_step(const Bytes& B, u32 p, u32 L, u32 w, u32& nxt, u64& vals, u32& info) {
    if (KIND == KIND_SNAPPY) {
        u32 x[2] = {0, 0};
        for (u32 k = 0; k < 8 && p + k < L; k++) x[k >> 2] |= B(p + k) << (8 * (k & 3));
        u32 v;
        snappy_step_w32(x[0], x[1], p, L, nxt, v, info);
        vals = v;
    } else rle_step(B, p, L, w, nxt, vals, info);
}

// Bytes k..k+7 / k..k+3 of a lane's register image w[] (its sub-tile plus 8 bytes of lookahead); k is a
// compile-time constant after unrolling, so these are plain register funnel shifts.
template <u32 N>
static __device__ __forceinline__ u64 win64(const u32 (&w)[N], int k) {
    const int d = k >> 2, sh = (k & 3) * 8;
    const u32 lo = sh ? (w[d] >> sh) | (w[d + 1] << (32 - sh)) : w[d];
    const u32 hi = sh ? (w[d + 1] >> sh) | (w[d + 2] << (32 - sh)) : w[d + 1];
    return lo | ((u64)hi << 32);
}
template <u32 N>
static __device__ __forceinline__ u32 win32(const u32 (&w)[N], int k) {
    const int d = k >> 2, sh = (k & 3) * 8;
    return sh ? (w[d] >> sh) | (w[d + 1] << (32 - sh)) : w[d];
}

template <int KIND, bool MARK, u32 NT>
__global__ __launch_bounds__(NT) void k_chain(const u8* __restrict__ src_base, const Region* __restrict__ regions,
                                                      const u32* __restrict__ tile_prefix, u32 n_regions,
                                                      ExitRec* __restrict__ exit_tbl, const TileEntry* __restrict__ tile_entry,
                                                      RunRec* __restrict__ run_tbl, u32* __restrict__ otile_first_run,
                                                      const PageDyn* __restrict__ page_dyn, const u32* __restrict__ tile_region) {
    static_assert(KIND != KIND_STRINGS, "strings have their own chain kernels");
    // NT threads share the tile: a thread's sub-tile is SUB = CH_T / NT bytes.  Two waves per tile halve every
    // thread's sweeps; the LDS memo (one entry per start) and with it the tiles resident per CU stay the same,
    // so twice the waves hide the latency of the LDS round trips.
    constexpr u32 SUB = CH_T / NT, WIN = SUB / 4 + 2, PITCH = NT + 1;
    static_assert(SUB == 16 || SUB == 32, "window loads below");
    auto memo_at = [](u32 rel) { return (rel % SUB) * PITCH + rel / SUB; };
    constexpr u32 E = CH_E_LEVELS;
    constexpr u32 OTK = KIND == KIND_SNAPPY ? SN_T : OT;  // output bytes per tile of the kernel that consumes the records
    constexpr u32 MEMO = SUB * PITCH;
    // memo per start: exit position, elements passed, and (levels) level values passed.  Levels keep
    // {exit, values} in one 8-byte entry so that a lookup is a single LDS read.
    __shared__ u64 m_tv[MEMO];
    __shared__ unsigned short m_cnt[MEMO];
    auto memo_get = [&](u32 mi, u32& tgt, u32& val) {
        const u64 e = m_tv[mi];
        tgt = (u32)e;
        val = (u32)(e >> 32);
    };
    auto memo_put = [&](u32 mi, u32 tgt, u32 val) {
        m_tv[mi] = (u64)tgt | ((u64)val << 32);
    };
    __shared__ u32 e_pos[MARK ? NT : 1], e_cnt[MARK ? NT : 1];
    __shared__ u64 e_val[MARK ? NT : 1];

    const u32 lane = threadIdx.x;
    const u32 tile = blockIdx.x;
    const u32 ri = tile_region ? tile_region[tile] : find_owner(tile_prefix, n_regions, tile);
    const Region R = regions[ri];
    const u32 L = R.len;
    const u32 tb = (tile - tile_prefix[ri]) * CH_T;  // region position of the tile start
    const u32 te = tb + CH_T;
    const u8* rsrc = src_base + R.src_off;

    TileEntry ent;
    if (MARK) {
        ent = tile_entry[tile];
        if (ent.pos == ERR_POS) return;  // the true chain never starts an element in this tile
        e_pos[lane] = ERR_POS;
    }

    // my sub-tile and 8 bytes of lookahead, in registers; bytes at or past L read as 0
    const u32 sub_lo = tb + lane * SUB, sub_hi = sub_lo + SUB;
    u32 w[WIN];
    if (sub_lo + 4 * WIN <= L) {
        // whole window inside the region: independent wide loads (any alignment)
        const u32x4 a = *(const u32x4_u*)(rsrc + sub_lo);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        if constexpr (SUB == 32) {
            const u32x4 b = *(const u32x4_u*)(rsrc + sub_lo + 16);
            w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        }
        const u64 c = *(const u64_u*)(rsrc + sub_lo + SUB);
        w[SUB / 4] = (u32)c; w[SUB / 4 + 1] = (u32)(c >> 32);
    } else {
#pragma unroll
        for (int d = 0; d < (int)WIN; d++) {
            const u32 p = sub_lo + 4 * d;
            u32 v = 0;
            if (p < L) {
                if (L - p >= 4) v = *(const u32_u*)(rsrc + p);
                else for (u32 k = 0; k < L - p; k++) v |= (u32)rsrc[p + k] << (8 * k);
            }
            w[d] = v;
        }
    }

    if (MARK && KIND == KIND_LEVELS && ent.n <= CH_SERIAL) {
        // ---- few runs start in this tile (long bit-packed or RLE runs, the common case): one lane walks
        // the true chain through an LDS copy of the tile, then one lane per run emits its record.
        __shared__ u32 r_out[CH_SERIAL + 1], r_info[CH_SERIAL], r_n;
        u32* const tb32 = (u32*)m_tv;  // the memo table is not used on this path
#pragma unroll
        for (int d = 0; d < (int)(SUB / 4); d++) tb32[lane * (SUB / 4) + d] = w[d];
        if (lane == NT - 1) { tb32[512] = w[SUB / 4]; tb32[513] = w[SUB / 4 + 1]; tb32[514] = 0; tb32[515] = 0; }
        __syncthreads();
        const u32 n_out = (u32)R.n_out;
        if (lane == 0) {
            u32 q = ent.pos, i = 0;
            u32 o = ent.vals < n_out ? (u32)ent.vals : n_out;
            while (i < ent.n && q < te && q < L) {
                const u32 rel = q - tb, di = rel >> 2, sh = (rel & 3) * 8;
                const u32 a = tb32[di], b = tb32[di + 1], c = tb32[di + 2], d3 = tb32[di + 3];
                const u32 xlo = __builtin_amdgcn_alignbit(b, a, sh), xhi = __builtin_amdgcn_alignbit(c, b, sh);
                const u32 b8 = __builtin_amdgcn_alignbit(d3, c, sh) & 0xFF;
                u32 nxt, vals, info;
                rle_step_w32(xlo, xhi, q, L, R.width, nxt, vals, info);
                if (!(info & 1) && R.width <= 8) {
                    const u32 d = (info >> 1) - q;  // the value byte sits 1..8 bytes after q
                    u32 val = 0;
                    if (R.width > 0 && nxt != ERR_POS)
                        val = d < 4 ? (xlo >> (8 * d)) & 0xFF : d < 8 ? (xhi >> (8 * (d - 4))) & 0xFF : b8;
                    info = val << 1;
                }
                if ((info & 1) && R.width == 0) info = 0;  // width 0 bit-packed groups are runs of zeros
                r_out[i] = o;
                r_info[i] = info;
                o = vals < n_out - o ? o + vals : n_out;
                i++;
                q = nxt;
            }
            r_out[i] = o;
            r_n = i;
        }
        __syncthreads();
        if (lane < r_n) {
            const u32 rank = ent.cnt + lane, o = r_out[lane], oe = r_out[lane + 1];
            RunRec rr;
            rr.out_off = o;
            rr.info = r_info[lane];
            run_tbl[R.run_base + rank] = rr;
            for (u32 t = (o + OT - 1) / OT; t * OT < oe; t++) otile_first_run[R.otile_base + t] = rank;  // tiles that start inside this run
        }
        return;
    }

    // ---- sweep 1: every start of my sub-tile learns where its chain leaves the sub-tile.  The marking
    // pass keeps each start's own step (next start, values, record) in registers for the replay below.
    u32 st_nxt[MARK ? SUB : 1], st_val[MARK ? SUB : 1], st_info[MARK ? SUB : 1];
#pragma unroll
    for (int k = (int)SUB - 1; k >= 0; k--) {
        const u32 p = sub_lo + k;
        const bool valid = p < L;
        u32 nxt, vals = 0;
        {
            u32 info;
            const u32 xlo = win32(w, k), xhi = win32(w, k + 4);
            elem_step_w32<KIND>(xlo, xhi, p, L, R.width, nxt, vals, info);
            if (MARK) {
                if (KIND == KIND_LEVELS && !(info & 1) && R.width <= 8) {
                    // RLE run of a narrow level: the record carries the value itself (width 0: value 0)
                    const u32 vp = info >> 1, d = vp - p;
                    u32 val = 0;
                    // the value byte sits 1..8 bytes after p: still inside the lane's 40-byte window
                    const u32 b8 = (w[(k + 8) >> 2] >> (8 * ((k + 8) & 3))) & 0xFF;
                    if (R.width > 0 && nxt != ERR_POS)
                        val = d < 4 ? (xlo >> (8 * d)) & 0xFF : d < 8 ? (xhi >> (8 * (d - 4))) & 0xFF : b8;
                    info = val << 1;
                }
                if (KIND == KIND_LEVELS && (info & 1) && R.width == 0) info = 0;  // width 0 bit-packed groups are runs of zeros
                st_val[k] = vals;
                st_info[k] = info;
            }
        }
        if (!valid) nxt = ERR_POS;
        if (MARK) st_nxt[k] = nxt;
        // a step that lands on a later start of this sub-tile takes over that start's memoised answer
        // (the lookup is unconditional: lanes without one read their own slot and drop the result)
        const u32 mi = (u32)k * PITCH + lane;
        const bool in = (nxt < sub_hi) & (nxt < L);
        const u32 mj = in ? (nxt - sub_lo) * PITCH + lane : mi;
        u32 t2, v2;
        memo_get(mj, t2, v2);
        const u32 c2 = m_cnt[mj];
        u32 tgt = in ? t2 : nxt;
        u32 c = in ? c2 + 1 : 1;                      // one run never holds more than 2^31 values
        u32 v = in ? sat_add32(vals, v2) : vals;
        if (!valid) { tgt = p; c = 0; v = 0; }
        memo_put(mi, tgt, v);
        m_cnt[mi] = (unsigned short)c;
    }
    __syncthreads();

    if (!MARK) {
        // ---- sweep 2, level A: sub-tile exits -> exits of groups of 8 sub-tiles (256 bytes).  Inside every
        // group, sub-tiles from last to first (all 8 groups at once, 4 starts per lane): a start whose
        // chain lands on a later start of its group takes over that start's already final answer.
        constexpr u32 GS = 256 / SUB;  // sub-tiles per group
        for (int j = (int)GS - 2; j >= 0; j--) {
#pragma unroll
            for (int i = 0; i < (int)(8 * SUB / NT); i++) {
                const u32 item = (u32)i * NT + lane, g = item / SUB, k = item % SUB;
                const u32 rel = (g * GS + (u32)j) * SUB + k;
                const u32 mi = k * PITCH + g * GS + (u32)j;
                u32 t, v;
                memo_get(mi, t, v);
                if (tb + rel < L && t < tb + (g + 1) * GS * SUB && t < L) {
                    const u32 mj = memo_at(t - tb);
                    u32 t2, v2;
                    memo_get(mj, t2, v2);
                    memo_put(mi, t2, sat_add32(v, v2));
                    m_cnt[mi] = (unsigned short)(m_cnt[mi] + m_cnt[mj]);
                }
            }
            __syncthreads();
        }
        // ---- level B: only the first E starts of the tile need their tile exit; each hops from group to
        // group (at most 7 hops) through the level-A answers and goes to the exit table.
        constexpr int PER = E / NT;
        u32 ht[PER], hc[PER], hv[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const u32 rel = (u32)i * NT + lane;  // E <= 256: all inside group 0
            const u32 mi = memo_at(rel);
            memo_get(mi, ht[i], hv[i]);
            hc[i] = m_cnt[mi];
        }
        for (int hop = 0; hop < 7; hop++) {
            bool any = false;
#pragma unroll
            for (int i = 0; i < PER; i++) {
                if (ht[i] < te && ht[i] < L) {
                    const u32 mj = memo_at(ht[i] - tb);
                    u32 t2, v2;
                    memo_get(mj, t2, v2);
                    hc[i] += m_cnt[mj];
                    hv[i] = sat_add32(hv[i], v2);
                    ht[i] = t2;
                    any = true;
                }
            }
            if (!__any(any)) break;
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            ExitRec x;
            x.tgt = ht[i];
            x.cnt = hc[i];
            x.vals = hv[i];
            exit_tbl[(u64)tile * E + (u32)i * NT + lane] = x;
        }
        return;
    }

    // ---- MARK: lane 0 follows the true chain from sub-tile to sub-tile
    if (lane == 0) {
        u32 q = ent.pos, c = ent.cnt;
        u64 v = ent.vals;
        while (q < te && q < L) {
            const u32 rel = q - tb, sidx = rel / SUB, mi = memo_at(rel);
            e_pos[sidx] = q; e_cnt[sidx] = c; e_val[sidx] = v;
            u32 t2, v2;
            memo_get(mi, t2, v2);
            c += m_cnt[mi];
            v += v2;
            q = t2;
        }
    }
    __syncthreads();
    // ---- every lane replays its sub-tile from its true entry: one record per true element
    u32 q = e_pos[lane];
    if (q == ERR_POS) return;
    u32 rank = e_cnt[lane];  // index of the element within the region
    // levels: index of the run's first level value, clamped to the page's value count (32-bit from here on)
    const u32 n_out = (u32)R.n_out;
    u32 o = e_val[lane] < n_out ? (u32)e_val[lane] : n_out;
    RunRec* const runs = run_tbl + R.run_base;
    u32* const ofr = otile_first_run + R.otile_base;
#pragma unroll
    for (int k = 0; k < (int)SUB; k++) {
        if (q == sub_lo + k && q < L) {
            const u32 vals = st_val[k];
            RunRec rr;
            rr.out_off = o;
            rr.info = st_info[k];
            runs[rank] = rr;
            const u32 oe = vals < n_out - o ? o + vals : n_out;
            // which output tiles start inside this run
            for (u32 t = (o + OTK - 1) / OTK; t * OTK < oe; t++) ofr[t] = rank;
            o = oe;
            rank++;
            q = st_nxt[k];
        }
    }
}

// =====================================================================================
// PLAIN BYTE_ARRAY chains.  A length prefix is cheap to parse and a wrong start dies fast (a random
// length overshoots the page), so strings do not need the every-byte sweep of k_chain:
//   k_str_exits   pass 1: the rows of a tile's exit table are its first CH_E_STRINGS plausible string starts (positions
//                 whose four bytes are a length that fits in the page), wherever they lie in the tile; for
//                 each, where its chain leaves the tile and how many strings it passes.  Same hop kernels
//                 as the level chains, with the row of an entry looked up through StrIndex.
//   k_str_mark    pass 2: up to 8 lanes walk the true chain from the tile's true entry and set one bit per
//                 string start; the wave then writes offsets[] (rank = prefix popcount) and the chars:
//                 every byte of the tile from the true entry on is a length byte or a char, chars are
//                 dense in the output, so they are compacted through LDS and leave as aligned 16-byte
//                 stores; the part of the tile's last string beyond the tile is copied by the whole wave.
// =====================================================================================
constexpr u32 STR_LDS_WORDS = CH_T / 4 + 4;
constexpr u32 STR_CP = 8;                       // checkpoints per chain: one per 256-byte block of the tile
constexpr u32 STR_CP_BYTES = CH_T / STR_CP;     // one every 256 bytes
static_assert(STR_CP * 2 == 16, "a checkpoint row is one 16-byte store");

// loads the lane's 32 bytes of the tile (zero past L) into w[0..7] and into the LDS image
static __device__ __forceinline__ void str_stage_tile(const u8* rsrc, u32 L, u32 tb, u32 lane, u32 (&w)[8], u32* s_tile) {
    const u32 sub_lo = tb + lane * CH_SUB;
    if (sub_lo + CH_SUB <= L) {
        const u32x4 a = *(const u32x4_u*)(rsrc + sub_lo), b = *(const u32x4_u*)(rsrc + sub_lo + 16);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    } else {
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const u32 p = sub_lo + 4 * d;
            u32 v = 0;
            if (p < L) {
                if (L - p >= 4) v = *(const u32_u*)(rsrc + p);
                else for (u32 k = 0; k < L - p; k++) v |= (u32)rsrc[p + k] << (8 * k);
            }
            w[d] = v;
        }
    }
    *(u32x4*)(s_tile + lane * 8) = u32x4{w[0], w[1], w[2], w[3]};
    *(u32x4*)(s_tile + lane * 8 + 4) = u32x4{w[4], w[5], w[6], w[7]};
    if (lane == 0) {
        // 4 bytes of lookahead: a length prefix may start in the last 3 bytes of the tile
        const u32 p = tb + CH_T;
        u32 v = 0;
        if (p < L) {
            if (L - p >= 4) v = *(const u32_u*)(rsrc + p);
            else for (u32 k = 0; k < L - p; k++) v |= (u32)rsrc[p + k] << (8 * k);
        }
        s_tile[CH_T / 4] = v;
        s_tile[CH_T / 4 + 1] = 0;
    }
}
// the 4 bytes at tile offset rel (rel < CH_T) of the LDS image
static __device__ __forceinline__ u32 str_len_at(const u32* s_tile, u32 rel) {
    const u32 d = rel >> 2;
    return __builtin_amdgcn_alignbit(s_tile[d + 1], s_tile[d], (rel & 3) * 8);
}

// A position is a candidate (a plausible string start) when its four bytes are a length that fits into the rest of the
// region (r = L - 3 - position) and is below STR_CAND_MAX.  The second condition is what keeps contents with zero
// bytes (UTF-16 text: a zero behind every letter) from looking like lengths in a page of many megabytes.  A true start
// that it rejects, a string of 1 MiB or more, is an entry that no table holds: the hop kernels step over that one
// string themselves (about a microsecond each, one after the other per page; with a bound of 64 KiB a page of 118 MB of
// strings of 60-200 KB took 1.14 ms instead of 0.28 ms).
constexpr u32 STR_CAND_MAX = 1u << 20;
static __device__ __forceinline__ u32 str_cand_room(int r) { return r > 0 ? ((u32)r < STR_CAND_MAX ? (u32)r : STR_CAND_MAX) : 0; }

// Pass 1: STR_A tiles per wave, CH_E_STRINGS rows per tile.
//   candidates  the positions of a tile whose four bytes are a length that fits in the page.  Every true string
//               start is one; in text hardly anything else is (the two positions in front of a true start, whose
//               bytes end in the zero bytes of its length).  The rows of a tile's exit table are its first
//               CH_E_STRINGS candidates in position order, whatever their distance from the tile start: long strings
//               enter a tile anywhere, and the first true start of the tile has only the candidates in front of it
//               that the tail of the entering string holds.  Their positions go to str_cand (entry_slot looks a
//               position up there).  The first STR_PROBE positions of four tiles are searched at once, 16 lanes per
//               tile; only tiles that have fewer rows than that (long strings) are searched further, 512 positions
//               of one tile at a time.
//   quarters    a walk through a whole tile is a hundred or more dependent LDS round trips while the tile holds its
//               place in the LDS, and that, not instruction issue, bounds the kernel.  So every tile is cut into
//               STR_Q quarters; quarters 1.. get rows of their own (their first STR_E2 candidates) and all rows, the
//               tile's and the quarters', only walk to the end of their quarter.  A tile row then reaches the end of
//               the tile in at most STR_Q - 1 hops through the quarter rows (a landing that is no quarter row is
//               walked to the end of its quarter on the spot).
//   one step    every row's candidate takes one step.  Most rows are done with it: the two positions in front of a
//               true start jump far (their length has the true length in a high byte) or die.
//   walks       the others, packed into the lanes of the wave whatever their tile, walk their chain to the end of
//               their quarter through the LDS image: strings started on the way, where the chain leaves the quarter
//               (or ERR_POS: a length that does not fit), and, for pass 2, the chain's first position in the
//               quarter's second 256-byte block.  Rows that lie on the same chain repeat each other's steps, which
//               costs nothing: all walks of the wave are one loop.
constexpr u32 STR_A = 8;        // tiles per wave
constexpr u32 STR_ITER = 4;     // groups of STR_A tiles per workgroup
constexpr u32 STR_PROBE = 128;  // leading positions of a tile searched for its rows in the shared sweep
constexpr u32 STR_Q = 4, STR_QB = CH_T / STR_Q;  // quarters per tile, bytes per quarter
constexpr u32 STR_E2 = 4;       // rows of a quarter
constexpr u32 STR_QPROBE = 64;  // leading positions of a quarter searched for its rows
constexpr u32 STR_TP = STR_A * CH_E_STRINGS;            // (tile, row) pairs of a wave
constexpr u32 STR_QP = STR_A * (STR_Q - 1) * STR_E2;    // (tile, quarter, row) pairs
constexpr u32 STR_PAIRS = STR_TP + STR_QP;
static_assert(STR_A % 4 == 0, "the probe takes four tiles at a time");
static_assert(STR_TP <= CH_THREADS, "one lane per tile row");
static_assert(STR_PAIRS <= 256, "a pair index fits in 8 bits");
static_assert(STR_QB == 2 * STR_CP_BYTES, "a quarter is two checkpoint blocks");
static_assert(STR_E2 == 4, "the rows of a quarter are read as one 64-bit word");
static_assert(CH_E_STRINGS % 8 == 0, "a tile's candidate row is read in 16-byte pieces");
__global__ __launch_bounds__(CH_THREADS) void k_str_exits(const u8* __restrict__ src_base, const Region* __restrict__ regions,
                                                          const u32* __restrict__ tile_prefix, u32 n_regions, u32 n_tiles,
                                                          ExitRec* __restrict__ exit_tbl, u32x4* __restrict__ str_cp,
                                                          const u32* __restrict__ tile_region,
                                                          unsigned short* __restrict__ str_cand, unsigned short* __restrict__ str_ncand,
                                                          const u32* __restrict__ gate, const u32* __restrict__ gate_any) {
    static_assert(STR_CP == 8, "a checkpoint row is 8 x u16");
    if (gate_any && !*gate_any) return;  // behind the single pass: only the chunks it gave up are decoded here
    constexpr u32 E = CH_E_STRINGS, A = STR_A;
    __shared__ __attribute__((aligned(16))) u32 s_tile[A * STR_LDS_WORDS];
    __shared__ u32 s_tb[A], s_len[A], s_found[A];
    __shared__ u64 s_src[A];
    // per pair: tile rows first (t * E + row), then the quarter rows ((t * (Q - 1) + q - 1) * E2 + row behind STR_TP)
    __shared__ __attribute__((aligned(16))) unsigned short s_start[STR_PAIRS];  // candidate, tile offset; 0xFFFF = none
    __shared__ u32 s_exit[STR_PAIRS], s_queue[STR_PAIRS];                      // tile offset where the walk ended, or ERR_POS
    __shared__ unsigned short s_cnt[STR_PAIRS], s_cp1[STR_PAIRS];              // strings started; first position in the quarter's second block
    __shared__ __attribute__((aligned(16))) unsigned short s_cp[STR_TP][STR_CP];
    // A workgroup takes STR_ITER groups of STR_A tiles one after the other.  Where a tile lies (region index -> region ->
    // bytes) is two dependent loads before the tile itself can be loaded: those are made once for all groups, and the
    // tiles of the next group are loaded into registers while the current group is worked on.
    __shared__ u32 s_mtb[STR_ITER * A], s_mlen[STR_ITER * A];
    __shared__ u64 s_msrc[STR_ITER * A];
    static_assert(STR_ITER * A <= CH_THREADS, "one lane per tile of the workgroup for the region lookups");
    const u32 lane = threadIdx.x, tile_first = blockIdx.x * STR_ITER * A;
    {
        u32 tb = 0, len = 0;
        u64 so = 0;
        const u32 tile = tile_first + lane;
        if (lane < STR_ITER * A && tile < n_tiles) {
            const u32 ri = tile_region ? tile_region[tile] : find_owner(tile_prefix, n_regions, tile);
            const Region R = regions[ri];
            if (!gate || gate[R.otile_base]) { tb = (tile - tile_prefix[ri]) * CH_T; len = R.len; so = R.src_off; }
        }
        if (gate && !__ballot(len != 0)) return;  // one wave per workgroup: no tile of a chunk that was given up
        if (lane < STR_ITER * A) { s_mtb[lane] = tb; s_mlen[lane] = len; s_msrc[lane] = so; }  // absent tiles: length 0, nothing is plausible
    }
    __syncthreads();
    // every tile of group `it` complete, lookahead included: its loads can all be in flight at once
    auto group_whole = [&](u32 it) {
        if (it >= STR_ITER || tile_first + (it + 1) * A > n_tiles) return false;
        bool w = true;
        for (u32 t = 0; t < A; t++) w = w && s_mtb[it * A + t] + CH_T + 4 <= s_mlen[it * A + t];
        return w;
    };
    u32x4 pa[A], pb[A];
    u32 pahead = 0;
#pragma unroll
    for (u32 t = 0; t < A; t++) pa[t] = pb[t] = u32x4{0, 0, 0, 0};
    auto prefetch = [&](u32 it) {
#pragma unroll
        for (u32 t = 0; t < A; t++) {
            const u8* p = src_base + s_msrc[it * A + t] + s_mtb[it * A + t] + lane * CH_SUB;
            pa[t] = *(const u32x4_u*)p;
            pb[t] = *(const u32x4_u*)(p + 16);
        }
        if (lane < A) pahead = *(const u32_u*)(src_base + s_msrc[it * A + lane] + s_mtb[it * A + lane] + CH_T);
    };
    bool have = group_whole(0);
    if (have) prefetch(0);
    for (u32 iter = 0; iter < STR_ITER; iter++) {
    const u32 t0 = tile_first + iter * A;
    if (t0 >= n_tiles) break;  // uniform
    const u32 nt = n_tiles - t0 < A ? n_tiles - t0 : A;
#pragma unroll
    for (u32 i = lane; i < STR_PAIRS; i += CH_THREADS) s_start[i] = 0xFFFF;
    if (lane < A) {
        s_tb[lane] = s_mtb[iter * A + lane]; s_len[lane] = s_mlen[iter * A + lane]; s_src[lane] = s_msrc[iter * A + lane];
        s_found[lane] = 0;
    }
    if (have) {
#pragma unroll
        for (u32 t = 0; t < A; t++) {
            *(u32x4*)(s_tile + t * STR_LDS_WORDS + lane * 8) = pa[t];
            *(u32x4*)(s_tile + t * STR_LDS_WORDS + lane * 8 + 4) = pb[t];
        }
        if (lane < A) { s_tile[lane * STR_LDS_WORDS + CH_T / 4] = pahead; s_tile[lane * STR_LDS_WORDS + CH_T / 4 + 1] = 0; }
    }
    __syncthreads();
    if (!have) {
        for (u32 t = 0; t < nt; t++) {
            u32 w[8];
            str_stage_tile(src_base + s_src[t], s_len[t], s_tb[t], lane, w, s_tile + t * STR_LDS_WORDS);
        }
    }
    const bool have_next = group_whole(iter + 1);
    if (have_next) prefetch(iter + 1);
    __syncthreads();
    // ---- candidates; a position p is one when len < min(max(L - 3 - p, 0), STR_CAND_MAX); str_step_win accepts every one
    // of them (and no p >= L is one).  First the leading STR_PROBE positions of four tiles at a time, 8 positions per lane.
#pragma unroll
    for (u32 it = 0; it < A / 4; it++) {
        const u32 t = it * 4 + (lane >> 4), j = lane & 15, o = j * 8;
        const u32* tl = s_tile + t * STR_LDS_WORDS + o / 4;
        const u32 x[3] = {tl[0], tl[1], tl[2]};
        const int room = (int)(s_len[t] - 3 - s_tb[t] - o);  // L - 3 - p for my first position
        u32 m = 0;
#pragma unroll
        for (u32 k = 0; k < 8; k++) {
            const u32 len = (k & 3) ? __builtin_amdgcn_alignbit(x[k / 4 + 1], x[k / 4], (k & 3) * 8) : x[k / 4];
            const int r = room - (int)k;
            m |= (u32)(len < str_cand_room(r)) << k;
        }
        const u32 mine = __popc(m);
        u32 incl = mine;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const u32 y = __shfl_up(incl, d);
            if (j >= (u32)d) incl += y;
        }
        u32 slot = incl - mine;
        const u32 total = __shfl(incl, lane | 15);
        while (m && slot < E) {
            const u32 k = __ffs(m) - 1;
            m &= m - 1;
            s_start[t * E + slot++] = (unsigned short)(o + k);
        }
        if (j == 0) s_found[t] = total;
    }
    __syncthreads();
    // tiles whose leading positions hold fewer candidates than rows: the rest of the tile, 512 positions at a time
    for (u32 t = 0; t < nt; t++) {
        const u32 tb = s_tb[t], L = s_len[t];
        const u32 rlim = tb + CH_T < L ? CH_T : L - tb;
        u32 found = s_found[t];
        if (found >= E || rlim <= STR_PROBE) continue;  // uniform
        for (u32 c0 = STR_PROBE; c0 < rlim && found < E; c0 += 512) {
            const u32 o = c0 + lane * 8;
            u32 m = 0;
            if (o < CH_T) {
                const u32* tl = s_tile + t * STR_LDS_WORDS + o / 4;
                const u32 x[3] = {tl[0], tl[1], tl[2]};
                const int room = (int)(L - 3 - tb - o);
#pragma unroll
                for (u32 k = 0; k < 8; k++) {
                    const u32 len = (k & 3) ? __builtin_amdgcn_alignbit(x[k / 4 + 1], x[k / 4], (k & 3) * 8) : x[k / 4];
                    const int r = room - (int)k;
                    m |= (u32)(len < str_cand_room(r)) << k;
                }
            }
            const u32 mine = __popc(m);
            u32 incl = mine;
            for (int d = 1; d < 64; d <<= 1) {
                const u32 y = __shfl_up(incl, d);
                if (lane >= (u32)d) incl += y;
            }
            u32 slot = found + incl - mine;
            found += __shfl(incl, 63);
            while (m && slot < E) {
                const u32 k = __ffs(m) - 1;
                m &= m - 1;
                s_start[t * E + slot++] = (unsigned short)(o + k);
            }
        }
        if (lane == 0) s_found[t] = found;
    }
    __syncthreads();
    // ---- rows of the quarters: their first STR_E2 candidates among their leading STR_QPROBE positions, 8 lanes each
    constexpr u32 QL = STR_QPROBE / 8;  // lanes per quarter
#pragma unroll
    for (u32 it = 0; it < (A * (STR_Q - 1) * QL + CH_THREADS - 1) / CH_THREADS; it++) {
        const u32 pi_raw = it * (CH_THREADS / QL) + lane / QL, j = lane % QL;  // pi = t * (Q - 1) + q - 1
        const bool pi_ok = pi_raw < A * (STR_Q - 1);
        const u32 pi = pi_ok ? pi_raw : 0;
        const u32 t = pi / (STR_Q - 1), o = (pi % (STR_Q - 1) + 1) * STR_QB + j * 8;
        const u32* tl = s_tile + t * STR_LDS_WORDS + o / 4;
        const u32 x[3] = {tl[0], tl[1], tl[2]};
        const int room = (int)(s_len[t] - 3 - s_tb[t] - o);
        u32 m = 0;
#pragma unroll
        for (u32 k = 0; k < 8; k++) {
            const u32 len = (k & 3) ? __builtin_amdgcn_alignbit(x[k / 4 + 1], x[k / 4], (k & 3) * 8) : x[k / 4];
            const int r = room - (int)k;
            m |= (u32)(len < str_cand_room(r)) << k;
        }
        const u32 mine = __popc(m);
        u32 incl = mine;
#pragma unroll
        for (int d = 1; d < (int)QL; d <<= 1) {
            const u32 y = __shfl_up(incl, d);
            if (j >= (u32)d) incl += y;
        }
        u32 slot = incl - mine;
        if (!pi_ok) m = 0;
        while (m && slot < STR_E2) {
            const u32 k = __ffs(m) - 1;
            m &= m - 1;
            s_start[STR_TP + pi * STR_E2 + slot++] = (unsigned short)(o + k);
        }
    }
    __syncthreads();
    // ---- one step per row, one lane per pair; the chains that stay inside their quarter are queued
    const u64 lt = (1ull << lane) - 1;
    u32 qn = 0;
#pragma unroll
    for (u32 i0 = 0; i0 < STR_PAIRS; i0 += CH_THREADS) {
        const u32 i = i0 + lane;
        const bool in = i < STR_PAIRS;
        const u32 t = i < STR_TP ? i / E : (i - STR_TP) / ((STR_Q - 1) * STR_E2);
        const u32 c0 = in ? s_start[i] : 0xFFFF;
        u32 rel = ERR_POS, cnt = 0;
        bool walk = false;
        if (c0 != 0xFFFF) {  // a candidate is plausible, so its step succeeds; absent tiles have no candidates
            const u32 tb = s_tb[t], L = s_len[t];
            const u32 rlim = tb + CH_T < L ? CH_T : L - tb;
            const u32 qe = (c0 / STR_QB + 1) * STR_QB, qend = qe < rlim ? qe : rlim;
            rel = c0 + 4 + str_len_at(s_tile + t * STR_LDS_WORDS, c0);
            cnt = 1;
            walk = rel < qend;
        }
        if (in) {
            s_exit[i] = rel;
            s_cnt[i] = (unsigned short)cnt;
            s_cp1[i] = 0xFFFF;
        }
        const u64 bal = __ballot(walk);
        if (walk) s_queue[qn + (u32)__popcll(bal & lt)] = i | (rel << 8);
        qn += (u32)__popcll(bal);
    }
    __syncthreads();
    // ---- walks to the end of the quarter, 64 chains at a time
    for (u32 base = 0; base < qn; base += CH_THREADS) {
        const bool have = base + lane < qn;
        const u32 e = have ? s_queue[base + lane] : 0;
        const u32 i = e & 0xFF;
        const u32 t = i < STR_TP ? i / E : (i - STR_TP) / ((STR_Q - 1) * STR_E2);
        const u32 tb = s_tb[t], L = s_len[t];
        const u32 rlim = tb + CH_T < L ? CH_T : L - tb;
        // a position needs its four length bytes inside the region: up to there the room is positive and
        // a length is accepted when len < room = L - 3 - position
        const u32 avail = L - tb, room0 = avail - 3, rlim4 = avail < 4 ? 0 : rlim < room0 ? rlim : room0;
        const u32 c0 = s_start[i];
        const u32 qe = (c0 / STR_QB + 1) * STR_QB, qend = qe < rlim ? qe : rlim, wend = qend < rlim4 ? qend : rlim4;
        const u32 odd_lo = qe - STR_CP_BYTES;                  // the quarter's second block
        const u32 thr = c0 < odd_lo ? odd_lo : 0xFFFFFFFFu;    // a row in the second block has no later block to report
        const u32* tl = s_tile + t * STR_LDS_WORDS;
        u32 rel = have ? e >> 8 : ERR_POS, cnt = 1, c1 = 0xFFFF;
        while (rel < wend) {  // ERR_POS ends the loop too
            if (rel >= thr && rel < c1) c1 = rel;  // positions only grow: the first one in the block stays
            const u32 len = str_len_at(tl, rel);  // two aligned words: an unaligned LDS read is slower
            rel = len < room0 - rel ? rel + 4 + len : ERR_POS;
            cnt++;
        }
        if (rel < qend) { rel = ERR_POS; cnt++; }  // a start in the last three bytes of the region
        if (have) {
            s_exit[i] = rel;
            s_cnt[i] = (unsigned short)cnt;
            s_cp1[i] = (unsigned short)c1;
        }
    }
    __syncthreads();
    // ---- tile rows: from quarter to quarter to the end of the tile; lane = (tile, row)
    if (lane < STR_TP) {
        const u32 t = lane / E;
        const u32 tb = s_tb[t], L = s_len[t];
        const u32 rlim = t < nt ? (tb + CH_T < L ? CH_T : L - tb) : 0;
        const u32 avail = L - tb, room0 = avail - 3, rlim4 = avail < 4 ? 0 : rlim < room0 ? rlim : room0;
        const u32* tl = s_tile + t * STR_LDS_WORDS;
        unsigned short* cp = s_cp[lane];
        *(u32x4*)cp = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        u32 pos = s_exit[lane], cnt = s_cnt[lane];
        {
            const u32 c1 = s_cp1[lane];
            if (c1 != 0xFFFF) cp[c1 / STR_CP_BYTES] = (unsigned short)c1;
        }
        while (pos < rlim) {  // pos lies in a later quarter than every position of the chain before it
            const u32 q = pos / STR_QB;  // >= 1
            cp[pos / STR_CP_BYTES] = (unsigned short)pos;
            const u32 rows0 = STR_TP + (t * (STR_Q - 1) + q - 1) * STR_E2;
            const u64 rows = *(const u64*)(s_start + rows0);
            u32 slot = STR_E2;
#pragma unroll
            for (u32 k = 0; k < STR_E2; k++)
                if ((u32)((rows >> (16 * k)) & 0xFFFF) == pos) slot = k;
            if (slot < STR_E2) {
                const u32 j = rows0 + slot, c1 = s_cp1[j];
                if (c1 != 0xFFFF) cp[c1 / STR_CP_BYTES] = (unsigned short)c1;
                cnt += s_cnt[j];
                pos = s_exit[j];
            } else {  // no row of the quarter: string by string to its end
                const u32 qe = (q + 1) * STR_QB, qend = qe < rlim ? qe : rlim, wend = qend < rlim4 ? qend : rlim4;
                u32 pb = pos / STR_CP_BYTES;
                while (pos < wend) {
                    const u32 len = str_len_at(tl, pos);
                    pos = len < room0 - pos ? pos + 4 + len : ERR_POS;
                    cnt++;
                    if (pos < qend && pos / STR_CP_BYTES != pb) { pb = pos / STR_CP_BYTES; cp[pb] = (unsigned short)pos; }
                }
                if (pos < qend) { pos = ERR_POS; cnt++; }
            }
        }
        if (t < nt) {
            StrExit x;
            x.tgt = pos == ERR_POS ? ERR_POS : tb + pos;
            x.cnt = cnt;
            ((StrExit*)exit_tbl)[(u64)t0 * E + lane] = x;
            str_cp[(u64)t0 * E + lane] = *(const u32x4*)cp;  // own row: no barrier needed
            str_cand[(u64)t0 * E + lane] = s_start[lane];
            if (lane % E == 0) { const u32 f = s_found[t]; str_ncand[t0 + t] = (unsigned short)(f < 0xFFFF ? f : 0xFFFF); }
        }
    }
    __syncthreads();  // the next group reuses the LDS
    have = have_next;
    }
}

// Pass 2 is a chain of dependent steps per tile (entry -> region -> tile bytes -> walk -> scan -> stores), so a
// wave takes STR_MG tiles and runs every step for all of them before it waits; the walks of all tiles share
// the same dependent LDS round trips (8 lanes per tile).  The LDS image of a tile is needed only for the walk
// and gives its place to the tile's compaction buffer afterwards.
constexpr u32 STR_MG = 2;
static_assert(CH_T + 16 <= STR_LDS_WORDS * 4, "the compaction buffer reuses the tile image");
static __device__ __forceinline__ void str_mark_tiles(u32 t0, const u8* __restrict__ src_base, const Region* __restrict__ regions,
                                                      const u32* __restrict__ tile_prefix, u32 n_regions, u32 n_tiles,
                                                      const TileEntry* __restrict__ tile_entry, const u32x4* __restrict__ str_cp,
                                                      const PageDyn* __restrict__ page_dyn, const u32* __restrict__ tile_region) {
    __shared__ __attribute__((aligned(16))) u32 s_buf[STR_MG][STR_LDS_WORDS];
    __shared__ u32 s_bits[STR_MG][64], s_last[STR_MG][2];
    const u32 lane = threadIdx.x;
    // loads that do not depend on each other are issued together: entries and region indices, then the regions,
    // the entries' checkpoint rows and the pages' offsets
    u32 epos[STR_MG], ecnt[STR_MG], eslot[STR_MG], ri[STR_MG];
    bool on[STR_MG];
#pragma unroll
    for (u32 t = 0; t < STR_MG; t++) {
        const bool in = t0 + t < n_tiles;
        const u32 tile = in ? t0 + t : t0;
        const TileEntry ent = tile_entry[tile];
        ri[t] = tile_region ? tile_region[tile] : find_owner(tile_prefix, n_regions, tile);
        epos[t] = ent.pos; ecnt[t] = ent.cnt; eslot[t] = ent.slot;
        on[t] = in && ent.pos != ERR_POS;  // ERR_POS: no string starts in this tile
    }
    bool any = false;
#pragma unroll
    for (u32 t = 0; t < STR_MG; t++) any = any || on[t];
    if (!any) return;
    // walker lanes: lane l < 8 * STR_MG serves block l % 8 of tile l / 8
    const u32 wt = lane / STR_CP, wj = lane % STR_CP;
    u32 w_epos = ERR_POS, w_tile = t0, w_slot = NO_SLOT;
#pragma unroll
    for (u32 t = 0; t < STR_MG; t++)
        if (wt == t && on[t]) { w_epos = epos[t]; w_tile = t0 + t; w_slot = eslot[t]; }
    const u32 w_b0 = (w_epos & (CH_T - 1)) / STR_CP_BYTES;  // the entry's block; tiles start at multiples of CH_T of the region
    unsigned short cp = 0xFFFF;
    if (w_epos != ERR_POS && w_slot != NO_SLOT) cp = ((const unsigned short*)(str_cp + (u64)w_tile * CH_E_STRINGS + w_slot))[wj];
    Region R[STR_MG];
    PageDyn pd[STR_MG];
    u32 tb[STR_MG], lim[STR_MG];
    u32 w[STR_MG][8];
#pragma unroll
    for (u32 t = 0; t < STR_MG; t++) {
        R[t] = regions[ri[t]];
        pd[t] = page_dyn[ri[t]];  // one entry per string page (region), from k_str_page_scan
    }
#pragma unroll
    for (u32 t = 0; t < STR_MG; t++) {
        const u32 tile = on[t] ? t0 + t : t0;
        tb[t] = (tile - tile_prefix[ri[t]]) * CH_T;
        lim[t] = tb[t] + CH_T < R[t].len ? tb[t] + CH_T : R[t].len;
        if (on[t]) str_stage_tile(src_base + R[t].src_off, R[t].len, tb[t], lane, w[t], s_buf[t]);
        s_bits[t][lane] = 0;
    }
    __syncthreads();
    // The true chain of a tile is walked by up to 8 lanes at once: pass 1 left, for the chain of every tabulated
    // entry, its first position inside each 256-byte block of the tile after the entry's; walker j takes block j.
    // An entry that pass 1 did not tabulate is walked by walker 0 alone.
    if (w_epos != ERR_POS) {
        u32 w_tb = 0, w_lim = 0, w_len = 0;
#pragma unroll
        for (u32 t = 0; t < STR_MG; t++)
            if (wt == t) { w_tb = tb[t]; w_lim = lim[t]; w_len = R[t].len; }
        const u32* tl = s_buf[wt];
        u32* bits = s_bits[wt];
        u32 q = ERR_POS, end = w_lim;
        if (w_slot == NO_SLOT) { if (wj == 0) q = w_epos; }
        else {
            if (wj == w_b0) q = w_epos;
            else if (wj > w_b0 && cp != 0xFFFF) q = w_tb + cp;
            const u32 b = w_tb + (wj + 1) * STR_CP_BYTES;
            end = b < w_lim ? b : w_lim;
        }
        // the walk in tile offsets.  A position needs its four length bytes inside the region: up to there the room is
        // positive and a length is accepted when len < room = L - 3 - position (what str_step_win accepts)
        const u32 avail = w_len - w_tb, room0 = avail - 3;
        const u32 rend = end - w_tb, rend4 = avail < 4 ? 0 : rend < room0 ? rend : room0;
        u32 rel = q == ERR_POS ? ERR_POS : q - w_tb;
        u32 last_rel = rel, last_nrel = ERR_POS;
        bool walked = false;
        while (rel < rend4) {
            atomicOr(&bits[rel >> 5], 1u << (rel & 31));
            const u32 len = str_len_at(tl, rel);
            last_rel = rel;
            rel = len < room0 - rel ? rel + 4 + len : ERR_POS;
            last_nrel = rel;
            walked = true;
        }
        if (rel < rend) {  // a start in the last three bytes of the region: its step fails
            atomicOr(&bits[rel >> 5], 1u << (rel & 31));
            last_rel = rel;
            rel = last_nrel = ERR_POS;
            walked = true;
        }
        // the walker that left the tile (or hit a bad length) holds the tile's last string
        if (walked && (rel == ERR_POS || w_tb + rel >= w_lim)) {
            s_last[wt][0] = w_tb + last_rel;
            s_last[wt][1] = last_nrel == ERR_POS ? ERR_POS : w_tb + last_nrel;
        }
    }
    __syncthreads();
    u32 n_chars[STR_MG], pad[STR_MG];
    u8* cdst[STR_MG];
#pragma unroll
    for (u32 t = 0; t < STR_MG; t++) {
        n_chars[t] = 0; pad[t] = 0; cdst[t] = nullptr;
        if (!on[t]) continue;
        const u32 sub_lo = tb[t] + lane * CH_SUB, sub_hi = sub_lo + CH_SUB;
        const u32 smask = s_bits[t][lane];  // true string starts of my sub-tile
        u32 prev = __shfl_up(smask, 1);     // every lane takes part: a masked-off lane would hand its neighbour 0
        if (lane == 0) prev = 0;
        const u32 lenmask = smask | (smask << 1) | (smask << 2) | (smask << 3) | ((prev >> 29) & 1) | (((prev >> 30) & 1) * 3) |
                            (((prev >> 31) & 1) * 7);
        const u32 k_lo = epos[t] <= sub_lo ? 0 : epos[t] >= sub_hi ? 32 : epos[t] - sub_lo;
        const u32 k_hi = lim[t] >= sub_hi ? 32 : lim[t] <= sub_lo ? 0 : lim[t] - sub_lo;
        u32 own = k_lo >= 32 ? 0 : 0xFFFFFFFFu << k_lo;
        if (k_hi < 32) own &= (1u << k_hi) - 1;
        const u32 charmask = own & ~lenmask;
        // one scan for both counts: chars in the low half, starts in the high half (each <= 2048)
        const u32 mine = __popc(charmask) | (__popc(smask) << 16);
        u32 incl = mine;
        for (int d = 1; d < 64; d <<= 1) {
            const u32 x = __shfl_up(incl, d);
            if (lane >= (u32)d) incl += x;
        }
        const u32 total = __shfl(incl, 63);
        n_chars[t] = total & 0xFFFF;
        const u32 pc = (incl - mine) & 0xFFFF, ps = (incl - mine) >> 16;
        const u32 cbase = (u32)pd[t].char_off;  // offsets are int32: a chunk holds less than 2^31 chars
        // offsets: string `rank` of the page starts at q, its chars at q + 4; every earlier string of the page
        // contributed 4 length bytes.  A page never holds more values than levels.
        {
            int32_t* const offs = (int32_t*)R[t].out + pd[t].val_off;
            const u32 n_out = (u32)R[t].n_out;
            u32 m = smask, rank = ecnt[t] + ps;
            while (m) {
                const u32 k = __ffs(m) - 1;
                m &= m - 1;
                if (rank < n_out) offs[rank] = (int32_t)(cbase + (sub_lo + k) - 4u * rank);
                rank++;
            }
        }
        // chars: char byte number i of the tile goes to chars[g0 + i]
        const u32 g0 = cbase + epos[t] - 4u * ecnt[t];
        cdst[t] = R[t].out2 + g0;
        pad[t] = (u32)((uintptr_t)cdst[t] & 15);
        u8* const stage = (u8*)s_buf[t];  // the walk is over: the tile image is dead
#pragma unroll
        for (int k = 0; k < (int)CH_SUB; k++)
            if ((charmask >> k) & 1) stage[pad[t] + pc + __popc(charmask & ((1u << k) - 1))] = (u8)(w[t][k >> 2] >> (8 * (k & 3)));
    }
    __syncthreads();
#pragma unroll
    for (u32 t = 0; t < STR_MG; t++) {
        if (!on[t]) continue;
        const u8* const stage = (const u8*)s_buf[t];
        u8* const cd = cdst[t];
        const u32 pd0 = pad[t], n_stg = pd0 + n_chars[t];
        // aligned 16-byte stores between the first and the last 16-byte boundary; the up to 15 bytes on either
        // side go one per lane (never a serial byte loop: every byte would cost a memory round trip)
        const u32 head_end = ((pd0 + 15) & ~15u) < n_stg ? ((pd0 + 15) & ~15u) : n_stg;
        const u32 tail_lo = (n_stg & ~15u) > head_end ? (n_stg & ~15u) : head_end;
        for (u32 c0 = head_end + lane * 16; c0 < tail_lo; c0 += 64 * 16) *(u32x4*)(cd - pd0 + c0) = *(const u32x4*)(stage + c0);
        if (lane < 16) { if (pd0 + lane < head_end) cd[lane] = stage[pd0 + lane]; }
        else if (lane < 32) { const u32 b = tail_lo + lane - 16; if (b < n_stg) cd[b - pd0] = stage[b]; }
        // the last string of the tile may run past the tile: its remaining chars follow directly
        const u32 lq = s_last[t][0], ln = s_last[t][1];
        if (ln != ERR_POS && ln > lim[t]) {
            const u32 s_lo = lq + 4 > lim[t] ? lq + 4 : lim[t];
            const u8* sp = src_base + R[t].src_off + s_lo;
            u8* dp = cd + n_chars[t];
            const u32 n = ln - s_lo, n16 = n & ~15u;
            for (u32 c0 = lane * 16; c0 < n16; c0 += 64 * 16) *(u32x4_u*)(dp + c0) = *(const u32x4_u*)(sp + c0);
            if (lane < 16 && n16 + lane < n) dp[n16 + lane] = sp[n16 + lane];
        }
    }
}
// Behind the single pass (gate_any given) the launch is a small grid whose workgroups loop over the tiles: nearly
// always no chunk was given up and every workgroup ends at once.
__global__ __launch_bounds__(CH_THREADS) void k_str_mark(const u8* __restrict__ src_base, const Region* __restrict__ regions,
                                                         const u32* __restrict__ tile_prefix, u32 n_regions, u32 n_tiles,
                                                         const TileEntry* __restrict__ tile_entry,
                                                         const u32x4* __restrict__ str_cp,
                                                         const PageDyn* __restrict__ page_dyn, const u32* __restrict__ tile_region,
                                                         const int* __restrict__ status, const u32* __restrict__ gate_any) {
    if (gate_any && !*gate_any) return;  // only tiles of the chunks that were given up have entries
    if (status && status[0]) return;  // an earlier kernel found the batch corrupt: value counts and offsets are not to be trusted
    for (u32 t0 = blockIdx.x * STR_MG; t0 < n_tiles; t0 += gridDim.x * STR_MG) {
        str_mark_tiles(t0, src_base, regions, tile_prefix, n_regions, n_tiles, tile_entry, str_cp, page_dyn, tile_region);
        __syncthreads();  // the next tiles reuse the LDS
    }
}

// =====================================================================================
// k_str_fused — the strings of a page batch in ONE pass over the page bytes (template_string.go:47-58,
// template_string_optional.go:68-79).  The two passes above read every page twice because a tile cannot know where
// the chain of length prefixes enters it.  Here every tile GUESSES its string starts from its own bytes, proves the
// guess against itself and against its neighbour, and writes offsets and chars at once; a chunk in which a proof fails
// is handed, on the device, to the exact two-pass kernels, which run behind this one and skip every other chunk.
//   guess      a position is a candidate when its four bytes are a plausible length (str_cand_room).  Every true start
//              is one.  In text the only others are the one or two positions in front of a true start (their length
//              has the true length in a higher byte), so the guess is: a candidate whose successor position is not a
//              candidate.
//   proof      inside the tile: every guessed start's end (start + 4 + length) is the next guessed start, and the last
//              one's end lies at or beyond the tile's end (its exit).  Between tiles: the exit of the tile before is this
//              tile's first guessed start; a page's first tile starts at 0.  By induction over the tiles of a page the
//              guessed starts are then exactly the chain the reference walks.  A tile without any candidate is passed
//              over by a string from before (its exit is its predecessor's).
//   ranks      the number of strings in front of a tile within its column chunk is a sum over the tiles before it: a
//              decoupled look-back over one 64-bit status word per tile {state, strings, exit}.  A tile publishes its
//              own count (AGG) as soon as its guess has proved itself, looks back over the words of the tiles before it
//              (64 at a time, one per lane) down to the first inclusive prefix, and publishes its inclusive prefix.
//              Chars and offsets of the whole chunk follow from the rank alone: string number g that starts at stream
//              position q has its chars at q - 4 g.
//   repair     a tile whose guess does not prove itself has published nothing: it waits for the exit of the tile
//              before it and walks its strings one by one from there (one lane; ~10 us, during which the tiles behind
//              it wait).  Not more than SF_REP_MAX tiles in a row may do so; then, and whenever a published guess
//              turns out wrong, a walk meets a bad length, or a wait runs out, the chunk is given up.
// Workgroups wait only for workgroups with lower block indices, which the dispatcher starts first; a wait that
// still runs out (SF_TIMEOUT) gives the chunk up instead of hanging.
// Nothing here reports corrupt data: a chunk that is given up gets its verdict from the two-pass kernels.
// =====================================================================================
#ifndef SF_EXP
#define SF_EXP 0
#endif
#ifndef SF_NT_STORE
#define SF_NT_STORE 1
#endif
// The guess uses the candidates of the two passes (a length below STR_CAND_MAX that fits in the page).  A bound of 64 KiB
// (one position in 65536 of random bytes a candidate instead of one in 4096), with candidates that claim an empty string
// dropped unless a candidate follows 4 bytes on (chars that begin with a zero byte make the position behind a true start
// a candidate too), was tried: contents of random bytes or UTF-16-like text still mislead the guess often enough for the
// chunk to be given up (a length of 256 or more in front of a zero byte cannot be told from a short one a byte later
// without following both), and strings of 64 KiB to 1 MiB, which the guess then does not see, were given up as well.
constexpr u32 SF_CAND_MAX = STR_CAND_MAX;
static_assert(SF_CAND_MAX % 65536 == 0, "sf_small_mask compares the upper half of the length");
static __device__ __forceinline__ u32 sf_cand_room(int r) { return str_cand_room(r); }
constexpr u64 SF_AGG = 1, SF_INCL = 2, SF_FAIL = 3;  // state of a unit's status word (0: nothing published yet)
constexpr u32 SF_PASS = 0xFFFFFFFDu;       // exit of a tile (unit) in which no string starts: the exit of the one before it
constexpr u32 SF_CNT_MAX = (1u << 28) - 1; // strings of a chunk that the status word can count
constexpr u64 SF_TIMEOUT = 2000000;        // wall-clock ticks (100 MHz) a unit waits for the units before it: 20 ms
constexpr u32 SF_THREADS = SF_WAVES * 64;
// status word of a unit: state (2 bits) | spill (2) | strings (28) | exit (32).  spill = bytes of the length prefix of the
// unit's last string that lie behind the unit's end (the next unit's first bytes are those, not chars)
static __device__ __forceinline__ u64 sf_pack(u64 state, u32 spill, u32 cnt, u32 exit) {
    return state << 62 | (u64)spill << 60 | (u64)cnt << 32 | exit;
}
static __device__ __forceinline__ u64 sf_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static __device__ __forceinline__ void sf_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Inclusive prefix sum over the 64 lanes of a wave in six DPP additions (shifts inside rows of 16 lanes, then the row
// totals broadcast upwards): no LDS crossbar round trips as with six shuffles.
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ u32 sf_dpp(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false); }
static __device__ __forceinline__ u32 sf_wave_scan(u32 v) {
    v += sf_dpp<0x111, 0xf>(v);  // row_shr:1
    v += sf_dpp<0x112, 0xf>(v);  // row_shr:2
    v += sf_dpp<0x114, 0xf>(v);  // row_shr:4
    v += sf_dpp<0x118, 0xf>(v);  // row_shr:8
    v += sf_dpp<0x142, 0xa>(v);  // row_bcast:15: lane 15 -> row 1, lane 47 -> row 3
    v += sf_dpp<0x143, 0xc>(v);  // row_bcast:31: lane 31 -> rows 2 and 3
    return v;
}
static __device__ __forceinline__ u32 sf_wave_sum(u32 v) { return (u32)__builtin_amdgcn_readlane((int)sf_wave_scan(v), 63); }
// Bit k of the result: the four bytes at byte k of the lane's 36-byte window W (nine little-endian words) are a number
// below SF_CAND_MAX, i.e. the 16-bit number at byte k + 2 is below SF_CAND_MAX / 65536.  Those lie in the halves of W's words (even k) and of
// the words shifted by one byte (odd k); one compare on a half word and one add-with-carry (c = 2 c + carry) per position.
static __device__ __forceinline__ u32 sf_small_mask(const u32 (&W)[9]) {
    u32 y[9];
#pragma unroll
    for (int d = 0; d < 8; d++) y[d] = __builtin_amdgcn_alignbit(W[d + 1], W[d], 8);  // bytes 4d+1 .. 4d+4
    y[8] = W[8] >> 8;                                                                // bytes 33 .. 35
    u32 c = 0;
    const u32 k16 = SF_CAND_MAX >> 16;
#pragma unroll
    for (int d = 7; d >= 0; d--)  // positions 4d+3, 4d+2, 4d+1, 4d: half words at bytes 4d+5, 4d+4, 4d+3, 4d+2
        asm("v_cmp_gt_u32_sdwa vcc, %[k], %[yn] src0_sel:DWORD src1_sel:WORD_0\n\t"
            "v_addc_co_u32_e32 %[c], vcc, %[c], %[c], vcc\n\t"
            "v_cmp_gt_u32_sdwa vcc, %[k], %[wn] src0_sel:DWORD src1_sel:WORD_0\n\t"
            "v_addc_co_u32_e32 %[c], vcc, %[c], %[c], vcc\n\t"
            "v_cmp_gt_u32_sdwa vcc, %[k], %[yd] src0_sel:DWORD src1_sel:WORD_1\n\t"
            "v_addc_co_u32_e32 %[c], vcc, %[c], %[c], vcc\n\t"
            "v_cmp_gt_u32_sdwa vcc, %[k], %[wd] src0_sel:DWORD src1_sel:WORD_1\n\t"
            "v_addc_co_u32_e32 %[c], vcc, %[c], %[c], vcc"
            : [c] "+v"(c)
            : [k] "v"(k16), [yn] "v"(y[d + 1]), [wn] "v"(W[d + 1]), [yd] "v"(y[d]), [wd] "v"(W[d])
            : "vcc");
    return c;
}

// What the units in front of unit `ui` say (one wave): strings in front of it within its chunk (units cf..ui-1), the
// exit of the nearest unit in front of it in its page that has one (0 for a page's first unit, rf), and the spill of the
// unit right in front of it.  False: the chunk was given up (by another unit, or here because the wait ran out).
static __device__ __forceinline__ bool sf_lookback(const u64* __restrict__ ustat, u32 ui, u32 cf, u32 rf, const u32* chunk_failed,
                                                   u32 lane, u32& excl, u32& pexit, u32& pspill) {
    // SF_LB blocks of 64 words are asked for per trip and gone through nearest first.  One block: with four, a batch of
    // two chunks (a thousand units of each in flight, the nearest inclusive prefix 100-200 units back) was slower, not
    // faster (0.67 against 0.59 ms; two blocks: 0.63): what a unit waits for there is the slowest of the units in front
    // of it to publish at all, and the more words it looks at, the more often one of them is late.  Asking for the
    // first block before the wave compacts its own tile, to look at it afterwards, gained nothing either: the units
    // right in front publish about when this one does, so the words have to be asked for again.
#ifndef SF_LB_N
#define SF_LB_N 1
#endif
    constexpr int SF_LB = SF_LB_N;
    excl = 0; pspill = 0;
    bool have_exit = ui == rf;
    pexit = 0;
    bool counted = false, first = true;
    long long base = ui;
    u64 t_start = 0;
    while (true) {
        u64 wds[SF_LB];
#pragma unroll
        for (int b = 0; b < SF_LB; b++) {
            const long long j = base - 1 - 64 * b - (long long)lane;
            wds[b] = j >= (long long)cf ? sf_load(ustat + j) : sf_pack(SF_INCL, 0, 0, 0);  // in front of the chunk: an empty prefix
        }
        bool again = false;
#pragma unroll
        for (int b = 0; b < SF_LB && !again; b++) {
            const u64 wd = wds[b];
            const long long j = base - 1 - (long long)lane;
            const u32 st = (u32)(wd >> 62);
            const u64 m_incl = __ballot(st == SF_INCL), m_nr = __ballot(st == 0);
            if (__ballot(st == SF_FAIL)) return false;
            const u32 fi = m_incl ? (u32)__ffsll((unsigned long long)m_incl) - 1 : 64;
            const u64 need = fi >= 63 ? ~0ull : (2ull << fi) - 1;  // lanes 0..fi
            if (m_nr & need) { again = true; break; }
            if (!counted) excl += sf_wave_sum(lane <= fi ? (u32)(wd >> 32) & SF_CNT_MAX : 0);
            if (first && ui > rf) pspill = (u32)((u64)__builtin_amdgcn_readfirstlane((int)(wd >> 32)) >> 28) & 3;  // lane 0: unit ui - 1, same page
            first = false;
            if (!have_exit) {
                const u64 mc = __ballot(lane <= fi && j >= (long long)rf && (u32)wd != SF_PASS);
                if (mc) {
                    pexit = (u32)__shfl(wd, (int)__ffsll((unsigned long long)mc) - 1);
                    have_exit = true;
                }
            }
            if (fi < 64) counted = true;
            if (counted && have_exit) return true;
            base -= 64;
            if (base <= (long long)cf && !have_exit) return false;  // cannot happen: a page's first unit has an exit
        }
        if (again) {
            if (__hip_atomic_load(chunk_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            const u64 now = wall_clock64();
            if (!t_start) t_start = now;
            else if (now - t_start > SF_TIMEOUT) return false;
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

// value of the lane above / below (0 where there is none): one DPP move, no trip through the LDS crossbar
static __device__ __forceinline__ u32 sf_lane_above(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }  // wave_shl:1
static __device__ __forceinline__ u32 sf_lane_below(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }  // wave_shr:1

// loads the lane's SF_SUB bytes of a tile that ends inside its page, or within four bytes of the page's end (bytes at or
// past L read as zero), into w and into the LDS image, with the four bytes behind the tile
static __device__ __forceinline__ void sf_stage_tail(const u8* rsrc, u32 L, u32 lane, u32 (&w)[SF_SUB / 4], u32* s_tile) {
    auto word_at = [&](u32 p) -> u32 {
        u32 v = 0;
        if (p < L) {
            if (L - p >= 4) v = *(const u32_u*)(rsrc + p);
            else for (u32 k = 0; k < L - p; k++) v |= (u32)rsrc[p + k] << (8 * k);
        }
        return v;
    };
#pragma unroll
    for (u32 d = 0; d < SF_SUB / 4; d++) {
        w[d] = word_at(lane * SF_SUB + 4 * d);
        s_tile[lane * (SF_SUB / 4) + d] = w[d];
    }
    if (lane == 0) {
        s_tile[SF_T / 4] = word_at(SF_T);  // a length prefix may start in the last 3 bytes of the tile
        s_tile[SF_T / 4 + 1] = 0;
    }
}

// One workgroup = one unit = SF_TPU tiles of SF_T bytes, a tile per wave, SF_SUB = 64 bytes per lane: what is done once per
// tile (the hand-over between lanes, the scan, the bookkeeping of ranks and exits, the set-up of the stores) is a third
// of the instructions, and this kernel is bound by instruction issue.
__global__ __launch_bounds__(SF_THREADS) void k_str_fused(const u8* __restrict__ src_base, const Region* __restrict__ regions,
                                                          const StrUnit* __restrict__ units,
                                                          u64* __restrict__ ustat, u32* __restrict__ chunk_failed,
                                                          u32* __restrict__ chunk_repairs,
                                                          u32* __restrict__ any_failed, u64* __restrict__ stamps) {
    constexpr u32 BAD = 0xFFFFFFFFu;  // s_cnt of a tile whose guess did not prove itself
    constexpr u32 NW = SF_SUB / 4;    // words per lane
    constexpr u32 NONE = 0xFFFFFFu;   // no start
    static_assert(SF_SUB == 64 && SF_TPU == SF_WAVES, "one 64-bit mask per lane, one tile per wave");
#define SF_STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[(u64)blockIdx.x * SF_STAMPS + (i)] = wall_clock64(); } while (0)
    SF_STAMP(0);
    const u64 cyc0 = stamps ? (u64)clock64() : 0;
    __shared__ __attribute__((aligned(16))) u32 s_buf[SF_TPU][SF_LDS_WORDS];
    __shared__ u32 s_cnt[SF_TPU], s_ent[SF_TPU], s_ext[SF_TPU], s_spill[SF_TPU], s_excl[SF_TPU], s_slot[SF_TPU];
    __shared__ u32 s_bits[SF_REP_MAX][SF_T / 32];
    __shared__ u32 s_alive, s_spin0;
    const u32 lane = threadIdx.x & 63, tu = threadIdx.x >> 6;  // the wave's tile of the unit
    // Workgroup b takes unit b of the launch order (a workgroup that took several units in turn, with the next unit's bytes
    // requested ahead, was tried: whatever hands the units out, one shared counter, eight of them or a fixed stride,
    // costs more in waits for units in front than the requests ahead save).  One table entry says where the unit's bytes
    // lie: the tiles are requested at once, the page's record comes beside them.
    const StrUnit U = units[blockIdx.x];
    const u32 ui = U.ui;
    if (stamps && threadIdx.x == 0) { if (ui == 0xFFFFFFF0u) stamps[1] = 1; SF_STAMP(10); }
    const u32 n_on = (U.rem + SF_T - 1) / SF_T < SF_TPU ? (U.rem + SF_T - 1) / SF_T : SF_TPU;  // the unit's tiles (>= 1)
    const Region R = regions[U.region];
    const bool on = tu < n_on;
    const u32 tb = U.tb0 + tu * SF_T;                 // page position of the tile
    const u32 rem = on ? U.rem - tu * SF_T : 0;       // bytes from the tile's start to the end of the page
    const u32 rlim = rem < SF_T ? rem : SF_T;         // bytes of the tile
    const u32 sub_lo = lane * SF_SUB;
    u32* const tl = s_buf[tu];
    u32 w[NW], nw = 0;  // the lane's bytes of the tile and the four bytes behind them
    const bool whole = rem >= SF_T + 4;
    if (whole) {
        // the tile lies in the page with four bytes behind it: all of the lane's bytes are requested before the first of
        // them is waited for
        const u8* const p = src_base + U.src_off + (u64)tu * SF_T + sub_lo;
        u32x4 x[NW / 4];
#pragma unroll
        for (u32 d = 0; d < NW / 4; d++) x[d] = *(const u32x4_u*)(p + 16 * d);
        nw = *(const u32_u*)(p + SF_SUB);
#pragma unroll
        for (u32 d = 0; d < NW / 4; d++) {
            w[4 * d] = x[d].x; w[4 * d + 1] = x[d].y; w[4 * d + 2] = x[d].z; w[4 * d + 3] = x[d].w;
            *(u32x4*)(tl + lane * NW + 4 * d) = x[d];
        }
        if (lane == 63) { tl[SF_T / 4] = nw; tl[SF_T / 4 + 1] = 0; }
    } else if (on) {
        sf_stage_tail(src_base + U.src_off + (u64)tu * SF_T, rem, lane, w, tl);
    } else {
#pragma unroll
        for (u32 d = 0; d < NW; d++) w[d] = 0;
    }
    __builtin_amdgcn_wave_barrier();  // a wave reads only the tile image it wrote itself until the first barrier
    // every load of the wave has arrived (its bytes are in the image).  Said out loud: the compiler does not see it on every
    // path, and a wait for loads that it puts further down also waits for the stores in flight by then (the status words),
    // a trip to memory each time.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    if (!whole) nw = on ? tl[lane * NW + NW] : 0;  // the next lane's first word, or the lookahead
    if (stamps && threadIdx.x == 0) { if (s_buf[0][0] == 0x12345678u) stamps[0] = 1; SF_STAMP(1); }  // the read waits for the tile's bytes

    // ---- the tile's starts S, the lanes' char bytes (every byte of the tile that is no length byte; the bytes in front of
    // its first start are the tail of the string that enters it) and the running sums of chars (low half) and starts (high
    // half) over the lanes.  The up to three length bytes that spill in from the tile before count as chars here.
    u64 S = 0, cmask = 0;
    u32 incl = 0;
    auto prep = [&]() {
        const u32 prev = sf_lane_below((u32)(S >> 32));  // the last three positions of the lane below hold at most one start
        const u64 lenmask = S | (S << 1) | (S << 2) | (S << 3) | ((prev >> 29) & 1) | (((prev >> 30) & 1) * 3) | (((prev >> 31) & 1) * 7);
        u64 own = ~0ull;
        if (rlim < SF_T) own = rlim >= sub_lo + SF_SUB ? ~0ull : rlim <= sub_lo ? 0ull : (1ull << (rlim - sub_lo)) - 1;
        cmask = own & ~lenmask;
        incl = sf_wave_scan((u32)__popcll(cmask) | ((u32)__popcll(S) << 16));
    };
    // ---- guess and proof inside the tile (positions relative to the tile)
    if (on) {
        const u32 la = (u32)__builtin_amdgcn_readlane((int)nw, 63);  // the four bytes behind the tile
        u64 C;
        if (rem >= SF_T + 4 + SF_CAND_MAX) {  // every position of the tile has SF_CAND_MAX bytes of page behind it
            const u32 W_lo[9] = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8]};
            const u32 W_hi[9] = {w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15], nw};
            C = sf_small_mask(W_lo) | ((u64)sf_small_mask(W_hi) << 32);
        } else {
            const int room0 = (int)rem - 3 - (int)sub_lo;
            u32 c2[2] = {0, 0};
#pragma unroll
            for (int k = 63; k >= 0; k--) {
                const u32 hi_w = k / 4 + 1 < (int)NW ? w[(k / 4 + 1) & 15] : nw;
                const u32 x = (k & 3) ? __builtin_amdgcn_alignbit(hi_w, w[k / 4], (k & 3) * 8) : w[k / 4];
                c2[k >> 5] = (c2[k >> 5] << 1) | (u32)(x < sf_cand_room(room0 - k));
            }
            C = c2[0] | ((u64)c2[1] << 32);
        }
        u32 c_next = sf_lane_above((u32)C) & 1;
        if (lane == 63) c_next = la < sf_cand_room((int)rem - 3 - (int)SF_T);
        const u64 Sm = C & ~((C >> 1) | ((u64)c_next << 63));
        // the first guessed start in the lanes above mine (nf), and of the tile (ent): the lanes that hold a start, as a
        // mask, say which lane to ask
        const u64 has = __ballot(Sm != 0);
        const u32 first = sub_lo + (u32)__ffsll((unsigned long long)Sm) - 1;
        const u64 above = (has >> 1) >> lane;
        const u32 nfl = __shfl(first, above ? (int)lane + (int)__ffsll((unsigned long long)above) : 0);
        const u32 nf = above ? nfl : NONE;
        const u32 ent = has ? (u32)__builtin_amdgcn_readlane((int)first, __ffsll((unsigned long long)has) - 1) : NONE;
        // the proof inside the tile.  A start with another start behind it in the same lane: its end is that one (lanes
        // with several starts go round together); the lane's last start: its end is the first start of the lanes above,
        // or, for the tile's last start, lies at or beyond the tile's end (the tile's exit)
        bool bad = false;
        u64 mi = Sm;
        while (true) {
            const u64 m2 = mi & (mi - 1);
            if (!__any(m2 != 0)) break;
            const u32 k = ((u32)__ffsll((unsigned long long)mi) - 1) & 63, k2 = (u32)__ffsll((unsigned long long)m2) - 1;
            const u32 len = str_len_at(tl, sub_lo + k);
            bad = bad | ((m2 != 0) & (k + 4 + len != k2));
            mi = m2 ? m2 : mi;
        }
        const u32 pl = sub_lo + ((63 - (u32)__clzll((long long)Sm)) & 63);  // the lane's last start (anything in a lane without one)
        const u32 nxl = pl + 4 + str_len_at(tl, pl);                       // a candidate's length fits in the page
        bad = bad | ((Sm != 0) & (above ? nxl != nf : nxl < rlim));
        const bool ok = !__any(bad);
        S = Sm;
        prep();
        const u32 cnt = (u32)__builtin_amdgcn_readlane((int)incl, 63) >> 16;
        u32 ext = SF_PASS, spill = 0;
        if (has) {
            const int l = 63 - (int)__clzll((unsigned long long)has);  // the lane of the tile's last start
            ext = tb + (u32)__builtin_amdgcn_readlane((int)nxl, l);
            const u32 lq = (u32)__builtin_amdgcn_readlane((int)pl, l);
            spill = lq + 4 > rlim ? lq + 4 - rlim : 0;
        }
        if (lane == 0) {
            s_cnt[tu] = ok ? cnt : BAD;
            s_ent[tu] = tb + ent;
            s_ext[tu] = ext;
            s_spill[tu] = spill;
            s_slot[tu] = BAD;
        }
    }
    SF_STAMP(2);
    __syncthreads();
    SF_STAMP(3);
    const bool all_ok = !__ballot(lane < n_on && s_cnt[lane < n_on ? lane : 0] == BAD);  // the same in every wave

    // ---- offsets and chars of the tile, in two steps.  Every byte of a tile that is no length byte is a char, so a tile
    // moves exactly its own bytes.  Step 1 needs the tile's starts and nothing from other tiles: the chars are compacted
    // into the tile's LDS image (the up to three length bytes that spill in from the tile before still count as chars and
    // lead them), the offsets, less what the rank adds, go behind them.  Step 2 needs the spill and the rank of the tile's
    // first string: string number g at stream position q has its chars at q - 4 g, so the tile's chars go to (stream
    // position of the tile) + spill - 4 (strings before it).
    u32 n_all = 0, a3 = 0;
    auto compact = [&]() {
        const u32 excl = incl - ((u32)__popcll(cmask) | ((u32)__popcll(S) << 16));
        n_all = (u32)__builtin_amdgcn_readlane((int)incl, 63) & 0xFFFF;
        const u32 pc = excl & 0xFFFF, ps = excl >> 16;
        // the chars keep their place within a 4-byte word of the destination whatever the rank turns out to be (it moves
        // them by a multiple of 4): byte i of the tile that is no length byte goes to byte a3 + i of the image
        a3 = (u32)((uintptr_t)R.out2 + R.run_base + tb) & 3;
        // from the lane's last byte down: shifting the mask out at the top gives the byte's bit as a carry, which both
        // moves the address down by one and decides the store (two instructions a byte); every read of the image is over
        u8* ap = (u8*)tl + (a3 + pc + (u32)__popcll(cmask));
#pragma unroll
        for (int h = 1; h >= 0; h--) {
            u32 mk = (u32)(cmask >> (32 * h));
#pragma unroll
            for (int k = 31; k >= 0; k--) {
                const bool c = __builtin_add_overflow(mk, mk, &mk);
                ap -= c;
                const u32 wd = w[8 * h + (k >> 2)];
                const u32 x = (k & 1) ? wd >> 8 : wd;  // bytes 1 and 3 of a word share one shift
                if (c && !(SF_EXP & 4)) *ap = (u8)((k & 2) ? x >> 16 : x);
            }
        }
        // Every start took four length bytes out of the chars, so behind them there is room for exactly its offset (at
        // most a3 + spill + alignment = 9 bytes more than the tile, and the image has 16).  From there the offsets are
        // stored in rank order, lane by lane.
        u32* ow = tl + ((a3 + n_all + 3) >> 2) + ps;
        u32 v = sub_lo - 4 * ps;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            u32 m = (u32)(S >> (32 * h));
            while (m) {
                *ow++ = v + (u32)__ffs(m) - 1;
                m &= m - 1;
                v -= 4;
            }
            v += 32;
        }
    };

    // ---- the unit: ranks, the proof between tiles, repairs (wave 0), while the other waves compact their chars
    if (tu == 0) {
        const bool mine = lane < n_on;
        u32 c = mine ? s_cnt[lane] : 0, e = mine ? s_ext[lane] : SF_PASS;
        const u32 en = mine ? s_ent[lane] : 0;
        const u32 t_lim = U.tb0 + ((lane + 1) * SF_T < U.rem ? (lane + 1) * SF_T : U.rem);
        const u32 cf = R.lv_base, rf = R.count_level;
        u32* const cfail = chunk_failed + R.otile_base;
        const u64 m_ex = __ballot(mine && e != SF_PASS);  // tiles with an exit of their own
        u32 tot = 0, uexit = SF_PASS;
        const u32 uspill0 = s_spill[n_on - 1];
        if (all_ok) {
            tot = sf_wave_sum(c);
            if (m_ex) uexit = __shfl(e, 63 - (int)__clzll((unsigned long long)m_ex));
            if (lane == 0) sf_store(ustat + ui, sf_pack(SF_AGG, uspill0, tot, uexit));
            compact();  // its own tile before it looks back: the units in front need the time anyway
        }
        u32 uexcl = 0, pexit = 0, pspill = 0;
        SF_STAMP(4);
        bool alive = sf_lookback(ustat, ui, cf, rf, cfail, lane, uexcl, pexit, pspill);
        SF_STAMP(5);
        u32 uspill = uspill0;
        if (alive && all_ok) {
            // the exit of the nearest tile in front of mine that has one (else the unit's predecessor's)
            const u64 below = m_ex & ((1ull << lane) - 1);
            const u32 e_below = __shfl(e, below ? 63 - (int)__clzll((unsigned long long)below) : 0);  // every lane takes part: a lane outside a shuffle is read as 0
            const u32 pe = below ? e_below : pexit;
            const bool wrong = mine && (c ? en != pe : pe < t_lim);
            if (__ballot(wrong)) alive = false;
            if (uexit == SF_PASS) uexit = pexit;
        } else if (alive) {
            // some guess failed: one lane goes through the tiles in order and walks the failed ones string by string
            if (lane == 0) {
                u32 pe = pexit, nrep = 0;
                for (u32 t = 0; t < n_on && alive; t++) {
                    const u32 tbt = U.tb0 + t * SF_T, remt = U.rem - t * SF_T, rend = remt < SF_T ? remt : SF_T, limt = tbt + rend;
                    const u32 ct = s_cnt[t];
                    if (ct != BAD) {
                        if (ct ? s_ent[t] != pe : pe < limt) { alive = false; break; }
                        if (ct) pe = s_ext[t];
                        continue;
                    }
                    if (nrep == SF_REP_MAX) { alive = false; break; }
                    u32 n = 0, spill = 0;
                    u32* const bits = s_bits[nrep];
                    for (u32 i = 0; i < SF_T / 32; i++) bits[i] = 0;
                    if (pe < limt) {
                        const u32* const tlt = s_buf[t];
                        const u32 room0 = remt - 3, rend4 = remt < 4 ? 0 : rend < room0 ? rend : room0;
                        u32 rel = pe - tbt, last_rel = rel;
                        while (rel < rend4) {
                            bits[rel >> 5] |= 1u << (rel & 31);
                            const u32 len = str_len_at(tlt, rel);
                            last_rel = rel;
                            rel = len < room0 - rel ? rel + 4 + len : ERR_POS;
                            n++;
                        }
                        if (rel < rend) rel = ERR_POS;  // a start in the last three bytes of the page
                        if (rel == ERR_POS) { alive = false; break; }  // a bad length: the two passes report it
                        spill = last_rel + 4 > rend ? last_rel + 4 - rend : 0;
                        pe = tbt + rel;
                        s_ext[t] = pe;
                    }
                    s_slot[t] = nrep++;
                    s_cnt[t] = n;
                    s_spill[t] = spill;
                }
                // A walk takes some 15 us during which every unit behind this one in the chunk waits: worth it for the odd
                // tile, not for contents that mislead the guess throughout (random bytes: 1.4 ms for 8 MB that the two
                // passes decode in 0.2 ms).  More than one walk per 32 units of the chunk so far, and four: given up.
                if (alive && nrep && atomicAdd(chunk_repairs + R.otile_base, nrep) + nrep > 4 + (ui - cf) / 32) alive = false;
                uexit = pe;
            }
            alive = __shfl((u32)alive, 0) != 0;
            uexit = __shfl(uexit, 0);
            if (alive) {
                c = mine ? s_cnt[lane] : 0;
                tot = sf_wave_sum(c);
                uspill = s_spill[n_on - 1];
            }
        }
        if (alive && (u64)uexcl + tot > SF_CNT_MAX) alive = false;
        if (alive) {
            const u32 ci = sf_wave_scan(c);
            if (mine) s_excl[lane] = uexcl + ci - c;
            if (lane == 0) { sf_store(ustat + ui, sf_pack(SF_INCL, uspill, uexcl + tot, uexit)); s_spin0 = pspill; }
        } else if (lane == 0) {
            __hip_atomic_store(cfail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(any_failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sf_store(ustat + ui, sf_pack(SF_FAIL, 0, 0, 0));
        }
        if (lane == 0) s_alive = alive;
        SF_STAMP(6);
    } else if (all_ok && on) {
        compact();
    }
    __syncthreads();
    if (!s_alive || !on) return;
    SF_STAMP(7);
    if (!all_ok) {  // after a repair: the starts of a walked tile come from the walk
        if (s_slot[tu] != BAD) {
            const u32* const bits = s_bits[s_slot[tu]];
            S = bits[2 * lane] | ((u64)bits[2 * lane + 1] << 32);
            prep();
        }
        compact();
    }
    __builtin_amdgcn_wave_barrier();
    SF_STAMP(8);
    const u32 excl = s_excl[tu];
    const u32 spin = tu == 0 ? s_spin0 : s_spill[tu - 1];
    const u64 q0 = R.run_base + tb + spin;  // stream position of the tile's first byte that is no spilled length byte
    if (q0 < 4ull * excl) {  // only a rank from a wrong guess elsewhere can do this: the chunk is being given up
        if (lane == 0) {
            __hip_atomic_store(chunk_failed + R.otile_base, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(any_failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    {
        int32_t* const offs = (int32_t*)R.out + excl;
        const u32 cap = R.width > excl ? R.width - excl : 0;  // entries of the chunk's offsets array in front of the closing one
        const u32 cnt = (u32)__builtin_amdgcn_readlane((int)incl, 63) >> 16;
        const u32 n = cnt < cap ? cnt : cap;
        const u32* const ow = tl + ((a3 + n_all + 3) >> 2);
        const u32 add = (u32)R.run_base + tb - 4u * excl;
#pragma unroll 1
        for (u32 i = lane; i < n; i += 64)
            if (!(SF_EXP & 1)) *as_global(offs + i) = (int32_t)(ow[i] + add);
    }
    u8* const cd = R.out2 + (q0 - 4ull * excl);
    // byte i of the tile's chars is byte a3 + spin + i of the image and goes to cd + i; pd0 % 4 == (a3 + spin) % 4
    const u32 pd0 = (u32)((uintptr_t)cd & 15);
    const int shift = (int)pd0 - (int)(a3 + spin);                 // a multiple of 4 between -4 and 12
    const u8* const stage = (const u8*)tl - shift;                 // stage[pd0 + i] = byte i of the tile's chars
    const u32 n_stg = pd0 + n_all - spin;
    // aligned 16-byte stores between the first and the last 16-byte boundary; the up to 15 bytes on either side go one per lane
    const u32 head_end = ((pd0 + 15) & ~15u) < n_stg ? ((pd0 + 15) & ~15u) : n_stg;
    const u32 tail_lo = (n_stg & ~15u) > head_end ? (n_stg & ~15u) : head_end;
#pragma unroll 1
    for (u32 c0 = head_end + lane * 16; c0 < tail_lo; c0 += 64 * 16) {
        const u32* const sp = (const u32*)(stage + c0);  // 4-byte aligned
#if SF_NT_STORE
        __builtin_nontemporal_store(u32x4{sp[0], sp[1], sp[2], sp[3]}, as_global((u32x4*)(cd - pd0 + c0)));
#else
        if (!(SF_EXP & 2)) *as_global((u32x4*)(cd - pd0 + c0)) = u32x4{sp[0], sp[1], sp[2], sp[3]};
#endif
    }
    if (lane < 16) { if (pd0 + lane < head_end) *as_global(cd + lane) = stage[pd0 + lane]; }
    else if (lane < 32) { const u32 b = tail_lo + lane - 16; if (b < n_stg) *as_global(cd + (b - pd0)) = stage[b]; }
    SF_STAMP(9);
    if (stamps && threadIdx.x == 0) stamps[(u64)blockIdx.x * SF_STAMPS + 15] = (u64)clock64() - cyc0;
#undef SF_STAMP
}

// k_str_reset — behind k_str_fused: the entry tables of the two passes start from 0xFF.  They are only looked at when a
// chunk was given up, so they are only filled then (a memset of the tables in front of every run moved 36 MB for nothing).
__global__ __launch_bounds__(256) void k_str_reset(u32x4* __restrict__ p, size_t n16, const u32* __restrict__ any_failed) {
    if (!*any_failed) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        p[i] = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
}

// k_str_totals — after k_str_fused, one thread per string page: the page's string count and its place in the chunk's
// arrays, from the inclusive prefixes of the tiles at the page's borders (what k_chain_resolve and k_str_page_scan
// leave on the two-pass path, which writes them itself for the chunks that were given up).
__global__ void k_str_totals(const Region* __restrict__ regions, u32 n_regions,
                             const u64* __restrict__ ustat, const u32* __restrict__ chunk_failed,
                             RegionTotal* __restrict__ totals, PageDyn* __restrict__ str_pd) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_regions) return;
    const Region R = regions[r];
    if (chunk_failed[R.otile_base]) return;
    // units of the page: [lo, hi); units of one chunk are numbered in page order from cf
    const u32 lo = R.count_level, hi = lo + (R.len + SF_UNIT - 1) / SF_UNIT, cf = R.lv_base;
    const u32 before = lo > cf ? (u32)(ustat[lo - 1] >> 32) & SF_CNT_MAX : 0;
    const u32 upto = hi > cf ? (u32)(ustat[hi - 1] >> 32) & SF_CNT_MAX : 0;
    RegionTotal T;
    T.cnt = upto - before; T.err = 0; T.vals = 0;
    totals[r] = T;
    PageDyn D;
    D.nn = T.cnt; D.pad = 0; D.val_off = before; D.char_off = R.run_base - 4ull * before;
    str_pd[r] = D;
}

// ---- hopping from tile to tile along the true chain, in three levels so that no thread makes more
// than a few dozen dependent loads:
//   k_chain_groups    per group of G tiles and per tabulated entry of its first tile: where the chain
//                     leaves the group (one lane per entry, <= G hops)
//   k_chain_resolve   one thread per region hops from group to group and records the group entries
//   k_chain_tiles     one thread per group hops through its tiles and records the tile entries

// Which row of an exit table belongs to an entry `off` bytes into the table's first tile (absolute tile index
// `tile`).  Levels and snappy elements: the tables hold the first E byte positions.  Strings: the tables hold
// the first E PLAUSIBLE string starts of the tile (positions whose four bytes are a length that fits in the
// page), so that an entry anywhere in the tile is tabulated, not just one in its first bytes: strings longer
// than a few dozen bytes enter a tile anywhere.  StrIndex::cand holds the rows' positions in ascending order:
// a position is looked up by comparing it with all of them (16 bytes per 8 rows).
// The tables of a group (LEVEL 1) and of a super-group (LEVEL 2) hold the first E plausible starts of ALL their
// tiles, so that a chain of strings longer than a tile, which enters most groups beyond their first tile, is
// still tabulated: a start's row there is its row in its own tile plus the plausible starts of the group's
// (super-group's) earlier tiles (StrIndex::gpre / spre, written by k_chain_groups / k_chain_supers).
// `first` = absolute index of the table's first tile, off = offset of the entry from that tile's start.
template <int KIND, int LEVEL>
static __device__ __forceinline__ u32 entry_slot(const StrIndex& sx, u32 first, u64 off) {
    constexpr u32 E = KIND == KIND_STRINGS ? CH_E_STRINGS : CH_E_LEVELS;
    if (KIND != KIND_STRINGS) return off < E ? (u32)off : NO_SLOT;
    const u32 tile = first + (u32)(off / CH_T), o = (u32)(off % CH_T);
    const u32x4* rows = (const u32x4*)(sx.cand + (u64)tile * E);  // the tile's rows: positions in ascending order, 0xFFFF = none
    u32x4 c[E / 8];
#pragma unroll
    for (u32 q = 0; q < E / 8; q++) c[q] = rows[q];
    const u32 pre = LEVEL == 0 ? 0 : LEVEL == 1 ? sx.gpre[tile] : sx.spre[tile];  // all loads in flight together
    u32 row = NO_SLOT;
#pragma unroll
    for (u32 q = 0; q < E / 8; q++) {
        const u32 wds[4] = {c[q].x, c[q].y, c[q].z, c[q].w};
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            if ((wds[k] & 0xFFFF) == o) row = q * 8 + 2 * k;
            if ((wds[k] >> 16) == o) row = q * 8 + 2 * k + 1;
        }
    }
    if (row == NO_SLOT) return row;
    return row + pre < E ? row + pre : NO_SLOT;
}
// Region position of row i of the table whose first tile is `first` (absolute index) and starts at region
// position base; the table spans n_scope tiles.  ERR_POS: the table has fewer rows.
template <int KIND, int LEVEL>
static __device__ __forceinline__ u32 slot_pos(const StrIndex& sx, u32 first, u32 base, u32 i, u32 n_scope) {
    if (KIND != KIND_STRINGS) return base + i;
    constexpr u32 G = chain_group_tiles(KIND_STRINGS);
    u32 t_lo = 0, acc = 0;
    if (LEVEL == 2) {  // find the group first: its total is the prefix of its last tile plus that tile's count
        u32 tot[CH_SUPER];  // loaded up front: uniform addresses, no dependent round trips
#pragma unroll
        for (u32 g = 0; g < CH_SUPER; g++) {
            const u32 last = first + ((g + 1) * G < n_scope ? (g + 1) * G : n_scope) - 1;
            tot[g] = g * G < n_scope ? (u32)sx.gpre[last] + sx.ncand[last] : 0;
        }
        u32 g_hit = CH_SUPER;
#pragma unroll
        for (u32 g = 0; g < CH_SUPER; g++) {
            if (g_hit == CH_SUPER && i < acc + tot[g]) g_hit = g;
            if (g_hit == CH_SUPER) acc += tot[g];
        }
        if (g_hit == CH_SUPER) return ERR_POS;
        t_lo = g_hit * G;
    }
    u32 n[G];
#pragma unroll
    for (u32 t = 0; t < G; t++) n[t] = t_lo + t < n_scope ? sx.ncand[first + t_lo + t] : 0;
#pragma unroll
    for (u32 t = 0; t < G; t++) {
        if (i < acc + n[t]) {
            const u32 c = sx.cand[(u64)(first + t_lo + t) * CH_E_STRINGS + (i - acc)];  // i - acc < n and i < 64
            return c == 0xFFFF ? ERR_POS : base + (t_lo + t) * CH_T + c;
        }
        acc += n[t];
    }
    return ERR_POS;
}

// One hop: the chain is at region position p (p < L); returns where it leaves p's tile.  With WALK false
// an entry that is not tabulated returns UNKNOWN_POS instead of being walked element by element.
template <int KIND, bool WALK = true>
static __device__ __forceinline__ u32 hop_tile(const Region& R, const GlobalBytes& B, const ExitRec* __restrict__ exit_tbl,
                                               const StrIndex& sx, u32 p, u32& cnt, u64& vals, u32* slot_out = nullptr) {
    constexpr u32 E = KIND == KIND_STRINGS ? CH_E_STRINGS : CH_E_LEVELS;
    const u32 L = R.len, t = p / CH_T, off = p % CH_T;
    const u32 slot = entry_slot<KIND, 0>(sx, R.tile_base + t, off);
    if (slot_out) *slot_out = slot;
    if (slot != NO_SLOT) {
        if (KIND == KIND_STRINGS) {  // string rows are {exit, strings passed}: 8 bytes
            const StrExit x = ((const StrExit*)exit_tbl)[(u64)(R.tile_base + t) * E + slot];
            cnt += x.cnt;
            return x.tgt;
        }
        const ExitRec x = exit_tbl[(u64)(R.tile_base + t) * E + slot];
        cnt += x.cnt; vals += x.vals;
        return x.tgt;
    }
    if (!WALK) return UNKNOWN_POS;
    // entry not tabulated: walk this tile element by element
    const u32 tend = (t + 1) * CH_T < L ? (t + 1) * CH_T : L;
    u32 q = p;
    while (q < tend) {
        u32 nxt; u64 v = 0;
        if (KIND == KIND_STRINGS) str_step(B, q, L, nxt);
        else { u32 info; elem_step<KIND>(B, q, L, R.width, nxt, v, info); }
        cnt++; vals += v;
        q = nxt;
    }
    return q;
}

// One hop over a group: the chain is at region position p; returns where it leaves p's group, through
// the group exit table when the entry is tabulated there, tile by tile otherwise.
template <int KIND>
static __device__ __forceinline__ u32 hop_group(const Region& R, const GlobalBytes& B, const ExitRec* __restrict__ exit_tbl,
                                                const ExitRec* __restrict__ gexit, const StrIndex& sx, u32 p, u32& cnt, u64& vals) {
    constexpr u32 E = KIND == KIND_STRINGS ? CH_E_STRINGS : CH_E_LEVELS;
    constexpr u32 G = chain_group_tiles(KIND);
    const u32 L = R.len, g = p / (G * CH_T), off = p - g * G * CH_T;
    ExitRec x;
    x.tgt = UNKNOWN_POS;
    const u32 slot = entry_slot<KIND, 1>(sx, R.tile_base + g * G, off);
    if (slot != NO_SLOT) x = gexit[(u64)(R.group_base + g) * E + slot];
    if (x.tgt != UNKNOWN_POS) {
        cnt += x.cnt; vals += x.vals;
        return x.tgt;
    }
    const u32 g_hi = (g + 1) * G * CH_T < L ? (g + 1) * G * CH_T : L;
    while (p < g_hi) p = hop_tile<KIND>(R, B, exit_tbl, sx, p, cnt, vals);
    return p;
}

template <int KIND>
__global__ __launch_bounds__(64) void k_chain_groups(const u8* __restrict__ src_base, const Region* __restrict__ regions,
                                                     const u32* __restrict__ group_prefix, u32 n_regions, u32 n_groups,
                                                     const ExitRec* __restrict__ exit_tbl, ExitRec* __restrict__ gexit,
                                                     const StrIndex sx, const u32* __restrict__ gate, const u32* __restrict__ gate_any) {
    constexpr u32 E = KIND == KIND_STRINGS ? CH_E_STRINGS : CH_E_LEVELS;
    constexpr u32 G = chain_group_tiles(KIND);
    static_assert(E % 64 == 0 || 64 % E == 0, "whole waves per group or whole groups per wave");
    if (gate_any && !*gate_any) return;
    // one lane per (group, row): E / 64 waves per group, or 64 / E groups per wave when a table has fewer rows than a wave has lanes
    constexpr u32 LPG = E < 64 ? E : 64;  // lanes of a wave that work on the same group
    const u32 gid = blockIdx.x * 64 + threadIdx.x;
    const u32 group = gid / E;
    const u32 i = gid % E;  // entry row of the group's first tile
    if (group >= n_groups) return;
    const u32 ri = find_owner(group_prefix, n_regions, group);
    const Region R = regions[ri];
    if (gate && !gate[R.otile_base]) return;
    const u32 L = R.len;
    const u32 g_lo = (group - group_prefix[ri]) * G * CH_T;
    const u32 g_hi = g_lo + G * CH_T < L ? g_lo + G * CH_T : L;
    const GlobalBytes B{src_base + R.src_off};
    const u32 t_first = (group - group_prefix[ri]) * G, n_rt = (L + CH_T - 1) / CH_T;
    const u32 n_scope = n_rt - t_first < G ? n_rt - t_first : G;
    if (KIND == KIND_STRINGS && i < LPG) {  // plausible starts in front of every tile of the group
        u32 n[G];
#pragma unroll
        for (u32 t = 0; t < G; t++) n[t] = t < n_scope ? sx.ncand[R.tile_base + t_first + t] : 0;  // same addresses for the group's lanes
        for (u32 tt = i; tt < n_scope; tt += LPG) {
            u32 acc = 0;
#pragma unroll
            for (u32 t = 0; t < G; t++)
                if (t < tt) acc += n[t];
            sx.gpre[R.tile_base + t_first + tt] = (unsigned short)(acc < 0xFFFF ? acc : 0xFFFF);
        }
    }
    u32 p = slot_pos<KIND, 1>(sx, R.tile_base + t_first, g_lo, i, n_scope), cnt = 0;
    u64 vals = 0;
    if (KIND != KIND_STRINGS && p >= g_hi) return;
    // Only one of the E candidate entries is the true one; a candidate whose chain lands on an entry that
    // some tile has not tabulated is given up (UNKNOWN_POS) rather than walked byte by byte: if it was
    // the true entry after all, k_chain_resolve hops that group tile by tile itself.
    if (p >= g_hi) p = UNKNOWN_POS;  // strings: a row without a candidate is never looked up
    while (p < g_hi) p = hop_tile<KIND, false>(R, B, exit_tbl, sx, p, cnt, vals);
    ExitRec x;
    x.tgt = p; x.cnt = cnt; x.vals = vals;
    gexit[(u64)group * E + i] = x;
}

// Super-groups (CH_SUPER groups each) add a fourth level for regions of thousands of tiles (string pages):
//   k_chain_supers        like k_chain_groups one level up: per super-group and tabulated entry, where the
//                         chain leaves the super-group (hops through the group exit table)
//   k_chain_resolve       then walks super-groups and records their entries
//   k_chain_super_groups  one thread per super-group walks its groups and records the group entries
template <int KIND>
__global__ __launch_bounds__(64) void k_chain_supers(const u8* __restrict__ src_base, const Region* __restrict__ regions,
                                                     const u32* __restrict__ super_prefix, u32 n_regions,
                                                     const ExitRec* __restrict__ gexit, ExitRec* __restrict__ sexit,
                                                     const StrIndex sx, const u32* __restrict__ gate, const u32* __restrict__ gate_any) {
    constexpr u32 E = KIND == KIND_STRINGS ? CH_E_STRINGS : CH_E_LEVELS;
    constexpr u32 G = chain_group_tiles(KIND);
    constexpr u32 GB = G * CH_T;  // bytes per group
    constexpr u32 BPG = E > 64 ? E / 64 : 1;
    if (gate_any && !*gate_any) return;
    const u32 sup = blockIdx.x / BPG;
    const u32 i = (blockIdx.x % BPG) * 64 + threadIdx.x;
    const u32 ri = find_owner(super_prefix, n_regions, sup);
    const Region R = regions[ri];
    if (gate && !gate[R.otile_base]) return;
    const u32 L = R.len;
    const u64 s_lo64 = (u64)(sup - super_prefix[ri]) * CH_SUPER * GB;
    const u32 s_lo = (u32)s_lo64;
    const u32 s_hi = s_lo64 + (u64)CH_SUPER * GB < L ? (u32)(s_lo64 + (u64)CH_SUPER * GB) : L;
    const u32 t_first = (sup - super_prefix[ri]) * CH_SUPER * G, n_rt = (L + CH_T - 1) / CH_T;
    const u32 n_scope = n_rt - t_first < CH_SUPER * G ? n_rt - t_first : CH_SUPER * G;
    if (KIND == KIND_STRINGS) {  // plausible starts in front of every tile of the super-group: groups before + inside the group
        u32 tot = 0;
        if (i * G < n_scope) {
            const u32 last = R.tile_base + t_first + ((i + 1) * G < n_scope ? (i + 1) * G : n_scope) - 1;
            tot = (u32)sx.gpre[last] + sx.ncand[last];
        }
        u32 incl = tot;
        for (int d = 1; d < 64; d <<= 1) {
            const u32 y = __shfl_up(incl, d);
            if (i >= (u32)d) incl += y;
        }
        const u32 excl = incl - tot;  // lane g: plausible starts in groups 0..g-1
        for (u32 k = i; k < CH_SUPER * G; k += 64) {
            const u32 before = __shfl(excl, k / G);
            if (k < n_scope) {
                const u32 v = before + sx.gpre[R.tile_base + t_first + k];
                sx.spre[R.tile_base + t_first + k] = (unsigned short)(v < 0xFFFF ? v : 0xFFFF);
            }
        }
    }
    if (i >= E) return;
    u32 p = slot_pos<KIND, 2>(sx, R.tile_base + t_first, s_lo, i, n_scope), cnt = 0;
    u64 vals = 0;
    if (KIND != KIND_STRINGS && p >= s_hi) return;
    if (p >= s_hi) p = UNKNOWN_POS;
    while (p < s_hi) {
        const u32 g = p / GB, off = p - g * GB;
        ExitRec x;
        x.tgt = UNKNOWN_POS;
        const u32 slot = entry_slot<KIND, 1>(sx, R.tile_base + g * G, off);
        if (slot != NO_SLOT) x = gexit[(u64)(R.group_base + g) * E + slot];
        if (x.tgt == UNKNOWN_POS) { p = UNKNOWN_POS; break; }  // not tabulated: the true chain is walked later
        cnt += x.cnt; vals += x.vals;
        p = x.tgt;
    }
    ExitRec x;
    x.tgt = p; x.cnt = cnt; x.vals = vals;
    sexit[(u64)sup * E + i] = x;
}

template <int KIND>
__global__ void k_chain_resolve(const u8* __restrict__ src_base, const Region* __restrict__ regions, u32 n_regions,
                                const ExitRec* __restrict__ exit_tbl, const ExitRec* __restrict__ gexit,
                                const u32* __restrict__ super_prefix, const ExitRec* __restrict__ sexit,
                                TileEntry* __restrict__ super_entry,
                                TileEntry* __restrict__ group_entry, RegionTotal* __restrict__ totals,
                                int* __restrict__ status, const StrIndex sx, const u32* __restrict__ gate) {
    constexpr u32 E = KIND == KIND_STRINGS ? CH_E_STRINGS : CH_E_LEVELS;
    constexpr u32 G = chain_group_tiles(KIND);
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_regions) return;
    const Region R = regions[r];
    if (gate && !gate[R.otile_base]) return;
    const u32 L = R.len;
    const GlobalBytes B{src_base + R.src_off};
    u32 p = 0, cnt = 0;
    u64 vals = 0;
    if (sexit) {
        const u64 SB = (u64)CH_SUPER * G * CH_T;  // bytes per super-group
        while (p < L) {
            const u32 sg = (u32)(p / SB);
            const u64 off = p - sg * SB;
            TileEntry se;
            se.pos = p; se.cnt = cnt; se.vals = vals; se.n = 0; se.slot = 0;
            super_entry[super_prefix[r] + sg] = se;
            ExitRec x;
            x.tgt = UNKNOWN_POS;
            const u32 slot = entry_slot<KIND, 2>(sx, R.tile_base + sg * CH_SUPER * G, off);
            if (slot != NO_SLOT) x = sexit[(u64)(super_prefix[r] + sg) * E + slot];
            if (x.tgt != UNKNOWN_POS) {
                cnt += x.cnt; vals += x.vals;
                p = x.tgt;
            } else {
                const u64 s_hi64 = (sg + 1) * SB;
                const u32 s_hi = s_hi64 < L ? (u32)s_hi64 : L;
                while (p < s_hi) p = hop_group<KIND>(R, B, exit_tbl, gexit, sx, p, cnt, vals);
            }
        }
    } else {
        while (p < L) {
            TileEntry ge;
            ge.pos = p; ge.cnt = cnt; ge.vals = vals; ge.n = 0; ge.slot = 0;
            group_entry[R.group_base + p / (G * CH_T)] = ge;
            p = hop_group<KIND>(R, B, exit_tbl, gexit, sx, p, cnt, vals);
        }
    }
    const bool err = p == ERR_POS;
    RegionTotal tot;
    tot.cnt = cnt; tot.err = err ? 1 : 0; tot.vals = vals;
    totals[r] = tot;
    if (err) set_status(status, ST_CORRUPT, r);
    else if (KIND == KIND_LEVELS && vals < R.n_out) set_status(status, ST_CORRUPT, r);  // defs[:NumValues] out of range
    else if (KIND == KIND_SNAPPY && vals != R.n_out) set_status(status, ST_CORRUPT, r);  // d != len(dst)
}

template <int KIND>
__global__ void k_chain_super_groups(const u8* __restrict__ src_base, const Region* __restrict__ regions,
                                     const u32* __restrict__ super_prefix, u32 n_regions, u32 n_supers,
                                     const ExitRec* __restrict__ exit_tbl, const ExitRec* __restrict__ gexit,
                                     const TileEntry* __restrict__ super_entry, TileEntry* __restrict__ group_entry,
                                     const StrIndex sx, const u32* __restrict__ gate_any) {
    constexpr u32 G = chain_group_tiles(KIND);
    if (gate_any && !*gate_any) return;
    const u32 sup = blockIdx.x * blockDim.x + threadIdx.x;
    if (sup >= n_supers) return;
    const TileEntry se = super_entry[sup];
    if (se.pos == ERR_POS) return;  // the true chain never starts an element in this super-group
    const u32 ri = find_owner(super_prefix, n_regions, sup);
    const Region R = regions[ri];
    const u32 L = R.len;
    const u64 SB = (u64)CH_SUPER * G * CH_T;
    const u64 s_hi64 = (u64)(sup - super_prefix[ri] + 1) * SB;
    const u32 s_hi = s_hi64 < L ? (u32)s_hi64 : L;
    const GlobalBytes B{src_base + R.src_off};
    u32 p = se.pos, cnt = se.cnt;
    u64 vals = se.vals;
    while (p < s_hi) {
        TileEntry ge;
        ge.pos = p; ge.cnt = cnt; ge.vals = vals; ge.n = 0; ge.slot = 0;
        group_entry[R.group_base + p / (G * CH_T)] = ge;
        p = hop_group<KIND>(R, B, exit_tbl, gexit, sx, p, cnt, vals);
    }
}

template <int KIND>
__global__ void k_chain_tiles(const u8* __restrict__ src_base, const Region* __restrict__ regions,
                              const u32* __restrict__ group_prefix, u32 n_regions, u32 n_groups,
                              const ExitRec* __restrict__ exit_tbl, const TileEntry* __restrict__ group_entry,
                              TileEntry* __restrict__ tile_entry, const StrIndex sx, const u32* __restrict__ gate_any) {
    constexpr u32 G = chain_group_tiles(KIND);
    if (gate_any && !*gate_any) return;
    const u32 group = blockIdx.x * blockDim.x + threadIdx.x;
    if (group >= n_groups) return;
    const TileEntry ge = group_entry[group];
    if (ge.pos == ERR_POS) return;  // the true chain never starts an element in this group
    const u32 ri = find_owner(group_prefix, n_regions, group);
    const Region R = regions[ri];
    const u32 L = R.len;
    const u32 g_lo = (group - group_prefix[ri]) * G * CH_T;
    const u32 g_hi = g_lo + G * CH_T < L ? g_lo + G * CH_T : L;
    const GlobalBytes B{src_base + R.src_off};
    u32 p = ge.pos, cnt = ge.cnt;
    u64 vals = ge.vals;
    while (p < g_hi) {
        TileEntry te;
        te.pos = p; te.cnt = cnt; te.vals = vals;
        const u32 idx = R.tile_base + p / CH_T;
        p = hop_tile<KIND>(R, B, exit_tbl, sx, p, cnt, vals, &te.slot);  // the entry's table row (NO_SLOT: not tabulated)
        te.n = cnt - te.cnt;
        tile_entry[idx] = te;
    }
}

// =====================================================================================
// k_levels_expand<2/4> — the standalone level decode with 16- or 32-bit results (widths up to 32; the page path
// and the one-byte standalone decode are k_levels in pqd_levels.hip).  One block per OT values: expands the runs
// that overlap the tile, found through the run table of the chain's marking pass.
// =====================================================================================
template <int ESZ>
__global__ __launch_bounds__(256) void k_levels_expand(const u8* __restrict__ src_base, const Region* __restrict__ regions,
                                                       const u32* __restrict__ otile_prefix, u32 n_regions,
                                                       const RunRec* __restrict__ run_tbl, const u32* __restrict__ otile_first_run,
                                                       const RegionTotal* __restrict__ totals, u32* __restrict__ otile_cnt) {
    __shared__ u32 s_out[EX_CAP + 1];
    __shared__ u32 s_info[EX_CAP];
    __shared__ __attribute__((aligned(16))) unsigned short s_mark[OT];
    __shared__ u32 s_wave[4];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 ot = blockIdx.x;
    const u32 ri = find_owner(otile_prefix, n_regions, ot);
    const Region R = regions[ri];
    const u32 L = R.len, w = R.width;
    const u8* rsrc = src_base + R.src_off;
    const u32 n_out = (u32)R.n_out;
    const u32 t_lo = (ot - otile_prefix[ri]) * OT;
    const u32 t_hi = t_lo + OT < n_out ? t_lo + OT : n_out;
    const u32 n_runs = totals[ri].cnt;
    u32 first = otile_first_run[ot];
    if (first == ERR_POS || totals[ri].err) return;  // short or corrupt block: status already set
    const RunRec* runs = run_tbl + R.run_base;
    // the runs this tile needs end with the run that owns the first value of the next tile
    u32 last = n_runs;  // exclusive
    if (ot + 1 < otile_prefix[ri + 1]) {
        const u32 nf = otile_first_run[ot + 1];
        if (nf != ERR_POS && nf + 1 < last) last = nf + 1;
    }

    const u32 o0 = t_lo + tid * 16;  // this thread's 16 values
    u64 acc_lo = 0, acc_hi = 0;      // ESZ == 1: the 16 result bytes
    u32 wide[ESZ == 1 ? 1 : 16];     // ESZ > 1: results (static indexing only)
    u32 cur = t_lo;                  // everything below `cur` is done
    for (;;) {
        // stage up to EX_CAP runs starting at `first`; they cover [cur, batch_hi)
        __syncthreads();
        const u32 m = last - first < EX_CAP ? last - first : EX_CAP;
        if (m == 0) break;
        for (u32 i = tid; i < m; i += 256) {
            const RunRec rr = runs[first + i];
            s_out[i] = rr.out_off;
            s_info[i] = rr.info;
        }
        if (tid == 0) s_out[m] = first + m < n_runs ? runs[first + m].out_off : n_out;
        // clear the marks (16 per thread)
        *(u32x4*)(s_mark + tid * 16) = u32x4{0, 0, 0, 0};
        *(u32x4*)(s_mark + tid * 16 + 8) = u32x4{0, 0, 0, 0};
        __syncthreads();
        const u32 batch_hi = s_out[m] < t_hi ? s_out[m] : t_hi;
        // every run that holds values marks the position where it starts (run 0 holds position `cur`);
        // runs without values never own a position, so no two runs mark the same one
        for (u32 i = tid; i < m; i += 256) {
            const u32 o = s_out[i], oe = s_out[i + 1];
            const u32 start = i == 0 ? cur : o;
            if (oe > start && start < batch_hi && start >= cur) s_mark[start - t_lo] = (unsigned short)(i + 1);
        }
        __syncthreads();
        // running maximum of the marks = (index + 1) of the run that owns each position
        u32 rk[16];
        {
            const u32x4 a = *(const u32x4*)(s_mark + tid * 16), b = *(const u32x4*)(s_mark + tid * 16 + 8);
            const u32 words[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            u32 run = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u32 mk = (words[k >> 1] >> (16 * (k & 1))) & 0xFFFF;
                run = mk > run ? mk : run;
                rk[k] = run;
            }
            u32 incl = run;
            for (int d = 1; d < 64; d <<= 1) {
                const u32 t = __shfl_up(incl, d);
                if (lane >= (u32)d && t > incl) incl = t;
            }
            if (lane == 63) s_wave[wv] = incl;
            u32 carry = __shfl_up(incl, 1);
            if (lane == 0) carry = 0;
            __syncthreads();
            for (u32 q = 0; q < wv; q++) carry = s_wave[q] > carry ? s_wave[q] : carry;
#pragma unroll
            for (int k = 0; k < 16; k++) rk[k] = rk[k] > carry ? rk[k] : carry;
        }
        // decode my positions inside [cur, batch_hi)
        u32 cur_r = 0, r_out = 0, r_info = 0, rle_val = 0, win_bit0 = 0;
        u64 win = 0;
        bool win_ok = false;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u32 pos = o0 + k;
            if (pos >= cur && pos < batch_hi) {
                const u32 r = rk[k];
                if (r != cur_r) {
                    cur_r = r;
                    r_out = s_out[r - 1];
                    r_info = s_info[r - 1];
                    win_ok = false;
                    if (!(r_info & 1)) {
                        if (w <= 8) rle_val = r_info >> 1;  // value stored in the record
                        else {
                            rle_val = 0;
                            const u32 bw = (w + 7) >> 3;
                            for (u32 b = 0; b < bw; b++) rle_val |= (u32)rsrc[(r_info >> 1) + b] << (8 * b);
                        }
                    }
                }
                u32 v;
                if (!(r_info & 1)) v = rle_val;
                else {
                    // bit-packed: value idx sits at bit idx*w of the packed bytes, LSB first
                    const u64 bit = (u64)(pos - r_out) * w;
                    if (!win_ok || bit < win_bit0 || bit - win_bit0 + w > 64) {
                        const u32 byte = (u32)(bit >> 3);
                        win = load_u64_guarded(rsrc, (r_info >> 1) + byte, L);
                        win_bit0 = byte * 8;
                        win_ok = true;
                    }
                    const u64 x = win >> (bit - win_bit0);
                    v = w == 32 ? (u32)x : (u32)(x & ((1ull << w) - 1));
                }
                if (ESZ == 1) {
                    if (k < 8) acc_lo |= (u64)(v & 0xFF) << (8 * k); else acc_hi |= (u64)(v & 0xFF) << (8 * (k - 8));
                } else wide[k] = v;
            }
        }
        if (batch_hi >= t_hi) break;
        // next batch starts with the run that owns position batch_hi
        first += m;
        cur = batch_hi;
    }

    // store + count
    u32 n_mine = 0;
    if (o0 < t_hi) n_mine = t_hi - o0 < 16 ? t_hi - o0 : 16;
    if (ESZ == 1) {
        u8* dst = R.out + o0;
        if (n_mine == 16) {
            u32x4 v;
            v.x = (u32)acc_lo; v.y = (u32)(acc_lo >> 32); v.z = (u32)acc_hi; v.w = (u32)(acc_hi >> 32);
            *(u32x4_u*)dst = v;
        } else {
            for (u32 k = 0; k < n_mine; k++) dst[k] = (u8)(k < 8 ? acc_lo >> (8 * k) : acc_hi >> (8 * (k - 8)));
        }
        if (R.count_level != ERR_POS) {
            // bytes equal to count_level: XOR, then detect zero bytes
            const u64 cl = (u64)(R.count_level & 0xFF) * 0x0101010101010101ull;
            const u64 keep_lo = n_mine >= 8 ? ~0ull : (n_mine ? (1ull << (8 * n_mine)) - 1 : 0);
            const u64 keep_hi = n_mine >= 16 ? ~0ull : (n_mine > 8 ? (1ull << (8 * (n_mine - 8))) - 1 : 0);
            auto zero_bytes = [](u64 x) {  // 0x80 in every byte of x that is zero
                return ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x) & 0x8080808080808080ull;
            };
            u32 c = __popcll(zero_bytes(acc_lo ^ cl) & keep_lo) + __popcll(zero_bytes(acc_hi ^ cl) & keep_hi);
            for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
            // one count per tile (no same-address atomics); the standalone decode passes no counter array
            __syncthreads();
            if (lane == 0) s_wave[wv] = c;
            __syncthreads();
            if (tid == 0) otile_cnt[ot] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        }
    } else if (ESZ == 2) {
        unsigned short* dst = (unsigned short*)R.out + o0;
#pragma unroll
        for (int k = 0; k < 16; k++) if ((u32)k < n_mine) dst[k] = (unsigned short)wide[k];
    } else {
        u32* dst = (u32*)R.out + o0;
#pragma unroll
        for (int k = 0; k < 16; k++) if ((u32)k < n_mine) dst[k] = wide[k];
    }
}


// =====================================================================================
// k_page_scan — one wave per chunk: running sums over its pages of the non-null counts (dense
// value index of each page's first value) and, for strings, of the char bytes; checks that
// bool/string pages hold exactly the bytes their values need.
// =====================================================================================
__global__ __launch_bounds__(64) void k_page_scan(const ChunkRec* __restrict__ chunks, u32 n_chunks,
                                                  const PageRec* __restrict__ pages, PageDyn* __restrict__ page_dyn,
                                                  ChunkDyn* __restrict__ chunk_dyn, int* __restrict__ status) {
    const u32 c = blockIdx.x;
    if (c >= n_chunks) return;
    const ChunkRec C = chunks[c];
    const u32 lane = threadIdx.x;
    u64 run_vals = 0, run_chars = 0;
    for (u32 base = 0; base < C.n_pages; base += 64) {
        const u32 pi = base + lane;
        u64 nn = 0, chars = 0;
        if (pi < C.n_pages) {
            const PageRec P = pages[C.first_page + pi];
            nn = C.max_def > 0 ? page_dyn[C.first_page + pi].nn : P.num_values;
            if (nn > P.num_values) {  // never more values than levels: every offset below depends on it
                set_status(status, ST_CORRUPT, C.first_page + pi);
                nn = P.num_values;
            }
            if (C.phys_type == PT_BYTE_ARRAY) {
                if (4 * nn > P.val_len) set_status(status, ST_CORRUPT, C.first_page + pi);
                else chars = P.val_len - 4 * nn;
            } else if (C.phys_type == PT_BOOLEAN) {
                const u64 need = (nn + 7) / 8;
                if (P.val_len < need) set_status(status, ST_CORRUPT, C.first_page + pi);
                else if (P.val_len > need && pi + 1 < C.n_pages) set_status(status, ST_NONCANONICAL, C.first_page + pi);
            }
        }
        // inclusive wave scans
        u64 sv = nn, sc = chars;
        for (int d = 1; d < 64; d <<= 1) {
            const u64 tv = __shfl_up(sv, d), tc = __shfl_up(sc, d);
            if (lane >= (u32)d) { sv += tv; sc += tc; }
        }
        if (pi < C.n_pages) {
            page_dyn[C.first_page + pi].val_off = run_vals + sv - nn;
            page_dyn[C.first_page + pi].char_off = run_chars + sc - chars;
            ((int64_t*)C.page_nonnull)[pi] = (int64_t)nn;  // the `sizes` slice of DoRead
        }
        run_vals += __shfl(sv, 63);
        run_chars += __shfl(sc, 63);
    }
    if (lane == 0) {
        // a required fixed-width Read takes pg.N values from the stream, however the pages split them
        if (C.max_def == 0 && C.width_bytes) run_vals = C.req_values;
        chunk_dyn[c].n_values = run_vals;
        chunk_dyn[c].chars = run_chars;
        if (C.max_def > 0 && C.max_rep == 0) chunk_dyn[c].n_rows = C.n_levels;  // not repeated: every level is a row
        if (C.max_def > 0) ((u32*)C.value_index)[(C.n_levels + 63) >> 6] = (u32)run_vals;  // the total closes the value index
        if (C.phys_type == PT_BYTE_ARRAY) {
            if (run_chars > 0x7fffffffull) set_status(status, ST_UNSUPPORTED, c);
            else ((int32_t*)C.offsets)[run_vals] = (int32_t)run_chars;
        }
        // binary.Read of n_values*w bytes from the concatenated values regions (io.ReadFull)
        if (C.width_bytes > 0 && run_vals * C.width_bytes > C.stream_len) set_status(status, ST_CORRUPT, c);
    }
}

// =====================================================================================
// k_value_index — the dense value index of the optional levels (valsFromDefs per level, fields.go:216-228; what
// the generated write functions do per row when they take `vals[0]` only for def == max, performance/parquet.go:104-113):
// value_index[w] = number of levels with def == max_def among the chunk's levels [0, 64 w), so the value of a
// non-null level i is values[value_index[i / 64] + popcount(validity bits of word i / 64 below bit i % 64)].
// One workgroup per page: it owns the 64-level words that START inside the page; the page's first value index
// comes from k_page_scan.  Entry ceil(n_levels / 64) (the total) is written by k_page_scan.
// =====================================================================================
// Workgroups of four waves, like those of the long kernels that run beside this one: a workgroup of sixteen waves needs
// sixteen free wave slots on one CU at once and waited a millisecond for them behind the string pass and the gather.
__global__ __launch_bounds__(256) void k_value_index(const ChunkRec* __restrict__ chunks, const PageRec* __restrict__ pages,
                                                     const PageDyn* __restrict__ page_dyn, const u32* __restrict__ page_ids,
                                                     const u64* __restrict__ page_lvl_off) {
    __shared__ u32 s_w[4], s_carry;
    const u32 pg = page_ids[blockIdx.x];
    const PageRec P = pages[pg];
    const ChunkRec C = chunks[P.chunk];
    const u64 l0 = page_lvl_off[pg], l1 = l0 + P.num_values;      // the page's levels within the chunk
    const u64 w0 = (l0 + 63) >> 6, w1 = (l1 + 63) >> 6;            // words that start inside the page
    if (w0 >= w1) return;
    const u64* const vb = (const u64*)C.validity;                 // 256-byte aligned, padded past its last word
    u32* const out = (u32*)C.value_index;
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) {
        // valid levels of the page in front of word w0: the upper part of word w0 - 1
        u32 head = 0;
        if (l0 & 63) head = __popcll(vb[w0 - 1] >> (l0 & 63));
        s_carry = (u32)page_dyn[pg].val_off + head;
    }
    __syncthreads();
    for (u64 base = w0; base < w1; base += 1024) {  // four consecutive words per thread
        const u64 w = base + tid * 4;
        u32 c[4];
#pragma unroll
        for (u32 i = 0; i < 4; i++) c[i] = w + i < w1 ? (u32)__popcll(as_global(vb)[w + i]) : 0;
        const u32 mine = c[0] + c[1] + c[2] + c[3];
        const u32 incl = sf_wave_scan(mine);
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        u32 before = s_carry, total = 0;
#pragma unroll
        for (u32 q = 0; q < 4; q++) {
            if (q < wv) before += s_w[q];
            total += s_w[q];
        }
        u32 run = before + incl - mine;
#pragma unroll
        for (u32 i = 0; i < 4; i++) {
            if (w + i < w1) *as_global(out + w + i) = run;
            run += c[i];
        }
        __syncthreads();
        if (tid == 0) s_carry += total;
    }
}

// =====================================================================================
// k_scatter — null-mask scatter: the dense values of a column with definition levels at their level slots.
// Slot i holds values[value index of i] when level i is non-null and 0 otherwise; the value index is
// value_index[i / 64] + popcount(validity bits of word i / 64 below bit i % 64).  OFFSETS (strings): the int32
// offsets at their slots, n_levels + 1 of them; a null slot takes the offset of the next value, so its string
// is empty.  One thread writes 16 bytes: 16 / W slots, all inside one validity word.
// =====================================================================================
template <int W, bool OFFSETS>
__global__ __launch_bounds__(256) void k_scatter(const u64* __restrict__ validity, const u32* __restrict__ value_index,
                                                 const u8* __restrict__ values, u8* __restrict__ dst, u64 n_slots, u64 n_words,
                                                 u32 n_src) {
    constexpr u32 PER = 16 / W;
    const u64 i0 = ((u64)blockIdx.x * 256 + threadIdx.x) * PER;
    if (i0 >= n_slots) return;
    const u64 wi = i0 >> 6;
    const u64 word = wi < n_words ? validity[wi] : 0;  // strings: the slot behind the last level may open a new word
    const u32 sh = (u32)(i0 & 63);
    u32 vi = value_index[wi] + (u32)__popcll(word & ((1ull << sh) - 1));
    const u32 bits = (u32)(word >> sh);
    u32 o[4] = {0, 0, 0, 0};
#pragma unroll
    for (u32 j = 0; j < PER; j++) {
        const bool on = (bits >> j) & 1;
        if (OFFSETS || on) {
            if (i0 + j < n_slots && vi < n_src) {  // n_src: entries the source array holds (a failed run may leave any index)
                if (W == 1) o[j >> 2] |= (u32)values[vi] << (8 * (j & 3));
                else if (W == 4) o[j] = ((const u32*)values)[vi];
                else { const u64 v = ((const u64*)values)[vi]; o[2 * j] = (u32)v; o[2 * j + 1] = (u32)(v >> 32); }
            }
        }
        vi += on ? 1u : 0u;
    }
    u8* const d = dst + i0 * W;
    if (i0 + PER <= n_slots) *(u32x4*)d = u32x4{o[0], o[1], o[2], o[3]};
    else for (u32 k = 0; k < (u32)(n_slots - i0) * W; k++) d[k] = (u8)(o[k >> 2] >> (8 * (k & 3)));
}

// =====================================================================================
// k_bools — PLAIN booleans: bit i%8 of byte i/8, LSB first, restarting at a byte boundary on every
// page (parquet.go:465-471).  One thread produces 16 output bytes.
// =====================================================================================
__global__ __launch_bounds__(256) void k_bools(const u8* __restrict__ src_base, const PageRec* __restrict__ pages,
                                               const u32* __restrict__ page_ids, const u32* __restrict__ btile_prefix,
                                               const u32* __restrict__ btile_owner, u32 n_bpages, const ChunkRec* __restrict__ chunks,
                                               const PageDyn* __restrict__ page_dyn, const int* __restrict__ status) {
    if (status[0]) return;  // an earlier kernel found the batch corrupt: value counts and offsets are not to be trusted
    const u32 bt = blockIdx.x;
    const u32 k = btile_owner[bt];  // which bool page the tile belongs to (a table: no search over the prefix)
    const u32 pg = page_ids[k];
    const PageRec P = pages[pg];
    const ChunkRec C = chunks[P.chunk];
    const PageDyn D = page_dyn[pg];
    const u64 count = C.max_def > 0 ? D.nn : P.num_values;
    const u64 i0 = (u64)(bt - btile_prefix[k]) * BT + threadIdx.x * 16;
    if (i0 >= count || (i0 >> 3) >= P.val_len) return;  // never past the page's values region (k_page_scan reports it)
    const u8* s = src_base + P.val_src_off + (i0 >> 3);
    const u32 bits = (u32)s[0] | (i0 + 8 < count && (i0 >> 3) + 1 < P.val_len ? (u32)s[1] << 8 : 0);
    u8* dst = C.values + D.val_off + i0;
    const u32 n = count - i0 < 16 ? (u32)(count - i0) : 16;
    if (n == 16) {
        u32x4 v;
        // spread 4 bits into 4 bytes: multiply trick on nibbles
        auto spread = [](u32 nib) { return ((nib & 1) | ((nib & 2) << 7) | ((nib & 4) << 14) | ((nib & 8) << 21)); };
        v.x = spread(bits & 15); v.y = spread((bits >> 4) & 15); v.z = spread((bits >> 8) & 15); v.w = spread((bits >> 12) & 15);
        *(u32x4_u*)dst = v;
    } else {
        for (u32 q = 0; q < n; q++) dst[q] = (bits >> q) & 1;
    }
}

// k_str_page_scan — one wave per BYTE_ARRAY chunk: where each page's strings go in the chunk's offsets and chars
// arrays, from the string counts the first pass found in the pages themselves (a page's strings are its non-null
// values, so the level decode would say the same; k_str_check compares the two afterwards).  The marking pass
// therefore needs nothing from the levels.  region_prefix[k] = first string region (page) of the k-th string chunk.
__global__ __launch_bounds__(64) void k_str_page_scan(const Region* __restrict__ regions, const u32* __restrict__ region_prefix,
                                                      u32 n_str_chunks, const RegionTotal* __restrict__ totals,
                                                      PageDyn* __restrict__ str_pd, const u32* __restrict__ gate) {
    const u32 c = blockIdx.x;
    if (c >= n_str_chunks) return;
    if (gate && !gate[c]) return;  // decoded by the single pass, which left this chunk's entries itself
    const u32 r0 = region_prefix[c], r1 = region_prefix[c + 1], lane = threadIdx.x;
    u64 run_vals = 0, run_chars = 0;
    for (u32 base = r0; base < r1; base += 64) {
        const u32 r = base + lane;
        u64 nn = 0, chars = 0;
        if (r < r1) {
            const Region R = regions[r];
            const RegionTotal T = totals[r];
            nn = T.err ? 0 : T.cnt;
            if (nn > R.n_out) nn = R.n_out;  // never more values than levels: the arrays are sized for that
            chars = 4 * nn <= R.len ? R.len - 4 * nn : 0;
        }
        u64 sv = nn, sc = chars;
        for (int d = 1; d < 64; d <<= 1) {
            const u64 tv = __shfl_up(sv, d), tc = __shfl_up(sc, d);
            if (lane >= (u32)d) { sv += tv; sc += tc; }
        }
        if (r < r1) {
            PageDyn D;
            D.nn = (u32)nn; D.pad = 0;
            D.val_off = run_vals + sv - nn;
            D.char_off = run_chars + sc - chars;
            str_pd[r] = D;
        }
        run_vals += __shfl(sv, 63);
        run_chars += __shfl(sc, 63);
    }
}

// Checks after the string chain: every page must hold exactly as many values as its levels say and
// end on its last byte.
__global__ void k_str_check(const Region* __restrict__ regions, u32 n_regions, const RegionTotal* __restrict__ totals,
                            const PageRec* __restrict__ pages, const ChunkRec* __restrict__ chunks,
                            const PageDyn* __restrict__ page_dyn, int* __restrict__ status) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_regions) return;
    const Region R = regions[r];
    const PageRec P = pages[R.page];
    const ChunkRec C = chunks[P.chunk];
    const u64 want = C.max_def > 0 ? page_dyn[R.page].nn : P.num_values;
    if (totals[r].err) return;
    if (totals[r].cnt < want) set_status(status, ST_CORRUPT, R.page);
    else if (totals[r].cnt > want) set_status(status, ST_NONCANONICAL, R.page);
}

// =====================================================================================
// k_snappy_ptr — decodes SN_T = 32 KiB of one snappy block's output per workgroup, in LDS, from the element
// records the chain kernels left (out_off, literal position or copy offset).
// Every output byte of the tile gets a 16-bit entry: either its value (a literal byte) or the tile position of
// the byte it copies (d[k] = d[k - offset], per byte, which also covers copies that overlap their own output).
// Rounds of x[i] = x[x[i]] then shorten every chain of copies of copies at once (after r rounds an entry has
// followed 2^r links), so the time depends on log2 of the dependency depth, not on the depth: a kernel that ran
// each copy when its source bytes were final took 0.5 us per link and 585 us per 64 KiB of run-like data.
// Entries are updated in place without barriers inside a round: whatever an entry holds at any time stands for
// the same byte value.  Last, 16 entries per thread are turned into 16 bytes and stored.
// snappy.Decode takes any offset up to the bytes written so far, so a copy may reach into (or continue from) an
// earlier tile; with the 64 KiB blocks of the usual encoders the odd tiles do.  Such a byte points at itself until
// the tiles that hold its source have raised their flags in `tile_done`; it is then fetched from the block's
// output in memory.  Workgroups take their tile from a ticket counter, so the tiles a workgroup waits for belong
// to workgroups that already run.  d < offset (a copy reaching before the start of the output) is ErrCorrupt.
// =====================================================================================
constexpr u32 SP_THREADS = 512;     // 8 waves per tile, two tiles per CU: the rounds are chains of dependent LDS reads and want waves to switch to
constexpr u32 SP_LIT = 0x8000u;       // entry = SP_LIT | byte value; otherwise a tile position
constexpr u32 SP_LONG_MAX = 512;      // literals of more than 64 bytes per tile
static_assert(SN_T <= 32768 && SN_T % (SP_THREADS * 16) == 0, "15-bit tile positions; whole rounds of 16 entries per thread");
static_assert(SN_T / 65 < SP_LONG_MAX, "every long literal of a tile has a slot");
static __device__ __forceinline__ u32 sp_byte(const u32x4& v, int k) {
    const u32 w = (k >> 2) == 0 ? v.x : (k >> 2) == 1 ? v.y : (k >> 2) == 2 ? v.z : v.w;
    return (w >> (8 * (k & 3))) & 0xFF;
}
// the 4 bytes of w as 4 value entries
static __device__ __forceinline__ u64 sp_lit4(u32 w) {
    const u32 lo = (w & 0xFF) | ((w & 0xFF00) << 8) | (SP_LIT | (SP_LIT << 16));
    const u32 hi = ((w >> 16) & 0xFF) | ((w >> 8) & 0xFF0000) | (SP_LIT | (SP_LIT << 16));
    return lo | ((u64)hi << 32);
}
// n (<= 64) value entries from the bytes of v at entry d (LDS takes stores at any alignment): whole groups of 4
// as one 8-byte store, the rest one by one
static __device__ __forceinline__ void sp_put_bytes(unsigned short* d, const u32x4 (&v)[4], u32 n) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const u32 wds[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32 k = 16 * c + 4 * q;
            if (k + 4 <= n) *(u64_u*)(d + k) = sp_lit4(wds[q]);
            else if (k < n) {
#pragma unroll
                for (int b = 0; b < 3; b++)
                    if (k + b < n) d[k + b] = (unsigned short)(SP_LIT | ((wds[q] >> (8 * b)) & 0xFF));
            }
        }
    }
}
__global__ __launch_bounds__(SP_THREADS) void k_snappy_ptr(const u8* __restrict__ src_base, const Region* __restrict__ regions,
                                                           const u32* __restrict__ otile_region, const RunRec* __restrict__ run_tbl,
                                                           const u32* __restrict__ otile_first_run,
                                                           const RegionTotal* __restrict__ totals, int* __restrict__ status,
                                                           u32* __restrict__ tile_done, u32* __restrict__ ticket, u32 n_otiles, u32 strict) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_snappy_ptr hands a tile's bytes to later tiles with device-scope stores, a wait for them (vmcnt) and a relaxed flag: that is ordered on gfx950; check it (or build with strict = 1 semantics) before compiling for another target"
#endif
    __shared__ __attribute__((aligned(16))) unsigned short s_x[SN_T];
    __shared__ u32 s_long[3 * SP_LONG_MAX];
    __shared__ u32 s_ticket, s_nlong, s_far_lo;
    const u32 tid = threadIdx.x;
    if (tid == 0) { s_ticket = atomicAdd(ticket, 1u); s_nlong = 0; s_far_lo = ERR_POS; }
    for (u32 i = tid; i < SN_T / 8; i += SP_THREADS) ((u32x4*)s_x)[i] = u32x4{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
    __syncthreads();
    const u32 ot = s_ticket;
    if (ot >= n_otiles) return;
    const u32 ri = otile_region[ot];
    const u32 first = otile_first_run[ot];
    const u32 nf = otile_first_run[ot + 1];
    const bool next_same = otile_region[ot + 1] == ri;
    const Region R = regions[ri];
    const RegionTotal T = totals[ri];
    if (first == ERR_POS || T.err || T.vals != R.n_out) return;  // corrupt block: status already set, no tile of it waits
    const u32 n_out = (u32)R.n_out;
    const u32 B = (ot - R.otile_base) * SN_T;
    const u32 t_hi = B + SN_T < n_out ? B + SN_T : n_out;
    const u32 n_tile = t_hi - B;
    const u8* rsrc = src_base + R.src_off;
    const RunRec* runs = run_tbl + R.run_base;
    u32 last = T.cnt;
    if (next_same && nf != ERR_POS && nf + 1 < last) last = nf + 1;

    // ---- 1. one element per thread: literal bytes and copy links.  The records of the next round are requested
    // before this round's are worked on (a record, then its literal bytes: two dependent trips to memory otherwise).
    auto fetch = [&](u32 e, RunRec& rr, u32& oe) {
        if (e < last) {
            rr = runs[e];
            oe = e + 1 < T.cnt ? runs[e + 1].out_off : n_out;
        }
    };
    RunRec rr_c = RunRec{0, 0}, rr_n = RunRec{0, 0};
    u32 oe_c = 0, oe_n = 0;
    fetch(first + tid, rr_c, oe_c);
    for (u32 base = first; base < last; base += SP_THREADS) {
        const u32 e = base + tid;
        fetch(e + SP_THREADS, rr_n, oe_n);
        const RunRec rr = rr_c;
        const u32 o = rr.out_off, oe = oe_c;
        rr_c = rr_n;
        oe_c = oe_n;
        if (e >= last) continue;
        const u32 lo = o > B ? o : B, hi = oe < t_hi ? oe : t_hi;  // an element may straddle tiles
        if (hi <= lo) continue;
        const u32 n = hi - lo, d0 = lo - B;
        if (!(rr.info & 1)) {
            const u32 sp = (rr.info >> 1) + (lo - o);
            if (n <= 64) {
                u32x4 v[4];
#pragma unroll
                for (int c = 0; c < 4; c++) v[c] = 16u * c < n ? load_u128_guarded(rsrc, sp + 16 * c, R.len) : u32x4{0, 0, 0, 0};
                sp_put_bytes(s_x + d0, v, n);
            } else {
                const u32 j = atomicAdd(&s_nlong, 1u);
                if (j < SP_LONG_MAX) { s_long[3 * j] = d0; s_long[3 * j + 1] = n; s_long[3 * j + 2] = sp; }
            }
        } else {
            const u32 off = rr.info >> 1;
            if (off > o || off == 0) set_status(status, ST_CORRUPT, ri);  // d < offset: the bytes stay zero
            else {
                // entry k copies tile position d0 + k - off; a source before the tile: the entry points at itself for
                // now.  Both are k plus a constant, four entries per store
                const bool has_far = lo - off < B;
                const u32 kf = has_far ? (B + off - lo < n ? B + off - lo : n) : 0;  // entries [0, kf) have their source before the tile
                for (u32 k = 0; k < n; k += 4) {
                    if (k + 4 <= n && (k >= kf || k + 4 <= kf)) {
                        const u32 c0 = (k >= kf ? d0 - off : d0) + k;
                        const u32 w0 = c0 * 0x00010001u + 0x00010000u, w1 = w0 + 0x00020002u;
                        *(u64_u*)(s_x + d0 + k) = w0 | ((u64)w1 << 32);
                    } else {
                        for (u32 b = k; b < k + 4 && b < n; b++) s_x[d0 + b] = (unsigned short)(b >= kf ? d0 + b - off : d0 + b);
                    }
                }
                if (has_far) atomicMin(&s_far_lo, lo - off);
            }
        }
    }
    __syncthreads();
    // long literals (incompressible stretches), by all threads
    {
        const u32 nl = s_nlong < SP_LONG_MAX ? s_nlong : SP_LONG_MAX;
        for (u32 j = 0; j < nl; j++) {
            const u32 d0 = s_long[3 * j], n = s_long[3 * j + 1], sp = s_long[3 * j + 2];
            for (u32 k = tid * 16; k < n; k += SP_THREADS * 16) {
                const u32x4 v = load_u128_guarded(rsrc, sp + k, R.len);
                if (((d0 + k) & 1) == 0 && k + 16 <= n) {
                    u32* const d = (u32*)(s_x + d0 + k);
#pragma unroll
                    for (int q = 0; q < 8; q++) d[q] = (SP_LIT | sp_byte(v, 2 * q)) | ((SP_LIT | sp_byte(v, 2 * q + 1)) << 16);
                } else {
#pragma unroll
                    for (int q = 0; q < 16; q++)
                        if (k + q < n) s_x[d0 + k + q] = (unsigned short)(SP_LIT | sp_byte(v, q));
                }
            }
        }
    }
    __syncthreads();

    // ---- 2. rounds of x[i] = x[x[i]], 8 entries per thread and step, until every entry is a value or points at a
    // byte that waits for an earlier tile
    const u32 n_grp = (n_tile + 7) / 8;
    u32 live = 0xFFFFFFFFu;  // bit t: my t-th group may still change (a group that has settled stays settled)
    static_assert(SN_T / 8 / SP_THREADS <= 32, "one bit per group of a thread");
    for (u32 round = 0; round < 20; round++) {  // 15 rounds settle any tile; the bound is a safety net
        bool unsettled = false;
        u32 t = 0;
        for (u32 g = tid; g < n_grp; g += SP_THREADS, t++) {
            if (!((live >> t) & 1)) continue;
            const u32x4 own = ((const u32x4*)s_x)[g];
            const u32 wds[4] = {own.x, own.y, own.z, own.w};
            u32 xs[8], ys[8];
#pragma unroll
            for (int k = 0; k < 8; k++) xs[k] = (wds[k >> 1] >> (16 * (k & 1))) & 0xFFFF;
#pragma unroll
            for (int k = 0; k < 8; k++) ys[k] = (xs[k] & SP_LIT) ? xs[k] : s_x[xs[k]];  // the reads of a group are in flight together
            bool g_live = false;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u32 x = xs[k], y = ys[k], i = g * 8 + k;
                if ((x & SP_LIT) || x == i) continue;
                if (y & SP_LIT) s_x[i] = (unsigned short)y;
                else if (y != x) { s_x[i] = (unsigned short)y; g_live = true; }
            }
            if (!g_live) live &= ~(1u << t);
            unsettled = unsettled || g_live;
        }
        if (!__syncthreads_or(unsettled)) break;
    }

    // ---- 3. bytes copied from earlier tiles of the block
    const u32 far_lo = s_far_lo;
    if (far_lo != ERR_POS) {
        const u32 q0 = far_lo / SN_T, q1 = B / SN_T;  // tiles q0 .. q1 - 1 of the block hold them
        bool ok = true;
        for (u32 q = q0 + tid; q < q1 && ok; q += SP_THREADS) {
            // a wait is bounded by time, not by polls: SN_WAIT_TICKS of the 100 MHz wall clock
            const u64 t_wait = wall_clock64();
            while (!(strict ? __hip_atomic_load(tile_done + R.otile_base + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                            : __hip_atomic_load(tile_done + R.otile_base + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (wall_clock64() - t_wait > SN_WAIT_TICKS) { ok = false; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            if (!ok && atomicCAS(status, 0, ST_INTERNAL) == 0) { status[1] = (int)ri; status[2] = (int)(R.otile_base + q); status[3] = (int)ot; }  // who waited for whom
        }
        __syncthreads();
        // The other tiles' bytes were stored with device-scope stores before their flags went up (below); device-scope
        // loads read them without a cache invalidate.
        fetch(first + tid, rr_c, oe_c);
        for (u32 base = first; base < last; base += SP_THREADS) {
            const u32 e = base + tid;
            fetch(e + SP_THREADS, rr_n, oe_n);
            const RunRec rr = rr_c;
            const u32 o = rr.out_off, oe = oe_c, off = rr.info >> 1;
            rr_c = rr_n;
            oe_c = oe_n;
            if (e >= last || !(rr.info & 1)) continue;
            const u32 lo = o > B ? o : B, hi = oe < t_hi ? oe : t_hi;
            if (hi <= lo || off > o || off == 0 || lo - off >= B) continue;
            const u32 stop = hi < B + off ? hi : B + off, n = stop - lo, d0 = lo - B, sp = lo - off;  // n <= 64
            // aligned words that cover [sp, sp + n): at most 17; all of them lie before this tile's first byte or in it,
            // inside the block's output (R.out is 16-byte aligned and padded to 16 bytes by its owner)
            const u32* const wp = (const u32*)(R.out + (sp & ~3u));
            const u32 mis = sp & 3, nw = (mis + n + 3) >> 2;
            u32 wd[17];
#pragma unroll
            for (int q = 0; q < 17; q++) wd[q] = (u32)q < nw ? __hip_atomic_load(wp + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            u32x4 v[4];
#pragma unroll
            for (int c = 0; c < 4; c++)
                v[c] = u32x4{__builtin_amdgcn_alignbit(wd[4 * c + 1], wd[4 * c], 8 * mis), __builtin_amdgcn_alignbit(wd[4 * c + 2], wd[4 * c + 1], 8 * mis),
                             __builtin_amdgcn_alignbit(wd[4 * c + 3], wd[4 * c + 2], 8 * mis), __builtin_amdgcn_alignbit(wd[4 * c + 4], wd[4 * c + 3], 8 * mis)};
            sp_put_bytes(s_x + d0, v, n);
        }
        __syncthreads();
    }

    // ---- 4. 16 entries -> 16 bytes (R.out is 16-byte aligned, B a multiple of the tile size)
    u8* const dst = R.out + B;
    for (u32 i = tid * 16; i < n_tile; i += SP_THREADS * 16) {
        const u32x4 a = *(const u32x4*)(s_x + i), b = *(const u32x4*)(s_x + i + 8);
        const u32 wds[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        u32 o4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            u32 x = (wds[k >> 1] >> (16 * (k & 1))) & 0xFFFF;
            if (!(x & SP_LIT)) x = s_x[x];
            o4[k >> 2] |= (x & 0xFF) << (8 * (k & 3));
        }
        // Device-scope stores: a later tile of the block, on whichever XCD it runs, may read these bytes once the
        // flag below is up.  (A device-scope release fence instead would write back the whole L2 once per tile:
        // measured 1.4 ms per 256 MB of output.)
        if (i + 16 <= n_tile) {
            __hip_atomic_store((u64*)(dst + i), o4[0] | ((u64)o4[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((u64*)(dst + i + 8), o4[2] | ((u64)o4[3] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (u32 k = i; k < n_tile; k++)
                __hip_atomic_store(dst + k, (u8)(o4[(k - i) >> 2] >> (8 * ((k - i) & 3))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // every thread waits for its own stores, then the flag goes up
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the stores above have been performed at device scope
    __syncthreads();
    // strict: a release store at device scope (writes the L2 back, slow) paired with acquire loads of the flag: the
    // textbook ordering, to compare with when a result is ever in doubt
    if (tid == 0) {
        if (strict) __hip_atomic_store(tile_done + ot, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(tile_done + ot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Level block sizes of pages whose payload only exists on the device (decompressed pages): the int32
// length in front of each hybrid-RLE block (rle.go:161-164), rep block first when the column repeats.
// sizes[2*i] / sizes[2*i+1] = rep / def block bytes including the 4 length bytes, 0 when absent,
// -1 when the length is negative or runs past the page.
__global__ void k_level_block_sizes(const u8* __restrict__ src, const PageRec* __restrict__ pages,
                                    const ChunkRec* __restrict__ chunks, u32 n_pages, int* __restrict__ sizes) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pages) return;
    const PageRec P = pages[i];  // val_src_off / val_len = the whole payload here
    const ChunkRec C = chunks[P.chunk];
    const u8* p = src + P.val_src_off;
    u64 l = 0;
    int out[2] = {0, 0};
    for (int lv = 0; lv < 2; lv++) {
        const bool present = lv == 0 ? C.max_rep > 0 : C.max_def > 0;
        if (!present) continue;
        if (l + 4 > P.val_len) { out[lv] = -1; break; }
        const int len = (int)((u32)p[l] | ((u32)p[l + 1] << 8) | ((u32)p[l + 2] << 16) | ((u32)p[l + 3] << 24));
        if (len < 0 || l + 4 + (u64)len > P.val_len) { out[lv] = -1; break; }
        out[lv] = len + 4;
        l += 4 + (u64)len;
    }
    sizes[2 * i] = out[0];
    sizes[2 * i + 1] = out[1];
}

// =====================================================================================
// Per-row offsets of repeated columns.  A row starts at every level with rep == 0; its value offset is the number
// of def == max_def before it.  Both come as bitmaps from the level decode (ChunkRec::row_starts, ::validity).
//   k_rows_count   per tile of RT levels: row starts and non-null counts (+ validity words)
//   k_rows_scan    one block per chunk: exclusive scan of the tile counts
//   k_rows_write   per tile: writes row_level_off / row_value_off
// =====================================================================================
// Row starts and non-null levels of every tile of RT levels: one wave per tile, 64 levels per lane from the bitmaps.
__global__ __launch_bounds__(256) void k_rows_count(const ChunkRec* __restrict__ chunks, const u32* __restrict__ rtile_prefix,
                                                    const u32* __restrict__ rtile_owner,
                                                    const u32* __restrict__ chunk_ids, u32 n_rchunks, u32 n_rtiles,
                                                    RowTile* __restrict__ tiles) {
    static_assert(RT == 64 * 64, "one 64-bit word of each bitmap per lane");
    const u32 lane = threadIdx.x & 63, rt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rt >= n_rtiles) return;
    const u32 k = rtile_owner[rt];  // which repeated chunk the tile belongs to (a table: no search over the prefix)
    const ChunkRec C = chunks[chunk_ids[k]];
    const u64 n = C.n_levels;
    const u64 i0 = (u64)(rt - rtile_prefix[k]) * RT + lane * 64;
    u64 valid = 0, starts = 0;
    if (i0 < n) {
        const u64 in_range = n - i0 >= 64 ? ~0ull : (1ull << (n - i0)) - 1;
        // the arrays are padded to whole 64-bit words, but not aligned to them: two 32-bit loads each
        const u32* pv = (const u32*)C.validity + (i0 >> 5);
        const u32* ps = (const u32*)C.row_starts + (i0 >> 5);
        valid = (pv[0] | ((u64)pv[1] << 32)) & in_range;
        starts = (ps[0] | ((u64)ps[1] << 32)) & in_range;
        if (i0 == 0) starts |= 1;  // the first level opens a row whatever its rep level (write_repeated.go:62-80)
    }
    u32 r = (u32)__popcll(starts), v = (u32)__popcll(valid);
    for (int d = 32; d > 0; d >>= 1) { r += __shfl_down(r, d); v += __shfl_down(v, d); }
    if (lane == 0) { tiles[rt].rows = r; tiles[rt].vals = v; }
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_rows(const ChunkRec* __restrict__ chunks, const u32* __restrict__ rtile_prefix,
                                              const u32* __restrict__ rtile_owner,
                                              const u32* __restrict__ chunk_ids, u32 n_rchunks,
                                              RowTile* __restrict__ tiles) {
    __shared__ u32 s_rows[4], s_vals[4];
    const u32 rt = blockIdx.x;
    const u32 k = rtile_owner[rt];  // which repeated chunk the tile belongs to (a table: no search over the prefix)
    const ChunkRec C = chunks[chunk_ids[k]];
    const u64 n = C.n_levels;
    const u64 i0 = (u64)(rt - rtile_prefix[k]) * RT + threadIdx.x * 16;
    u32 valid = 0, starts = 0;
    if (i0 < n) {
        const u32 in_range = n - i0 >= 16 ? 0xFFFFu : (1u << (n - i0)) - 1;
        // the level decode left both as bitmaps (16 levels = one 16-bit word; the arrays are padded)
        valid = as_global((const unsigned short*)C.validity)[i0 >> 4] & in_range;
        starts = as_global((const unsigned short*)C.row_starts)[i0 >> 4] & in_range;
        if (i0 == 0) starts |= 1;  // the first level opens a row whatever its rep level (write_repeated.go:62-80)
    }
    // block-wide exclusive scan of (row starts, valid): a wave holds 1024 levels, so both sums fit in one half word each
    // and go through one scan
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 r = __popc(starts), v = __popc(valid);
    const u32 both = sf_wave_scan(r | v << 16);
    const u32 sr = both & 0xFFFF, sv = both >> 16;
    if (lane == 63) { s_rows[wv] = sr; s_vals[wv] = sv; }
    __syncthreads();
    u32 br = 0, bv = 0, tot_r = 0, tot_v = 0;
    for (u32 q = 0; q < 4; q++) {
        if (q < wv) { br += s_rows[q]; bv += s_vals[q]; }
        tot_r += s_rows[q]; tot_v += s_vals[q];
    }
    if (!WRITE) {
        if (threadIdx.x == 0) { tiles[rt].rows = tot_r; tiles[rt].vals = tot_v; }
        return;
    }
    // rows of this tile, staged in LDS (tile-relative, 12 bits each) so that the 8-byte stores below are
    // coalesced: a thread's own row starts are strided in the output
    __shared__ unsigned short s_rl[WRITE ? RT : 1], s_rv[WRITE ? RT : 1];
    u32 lrow = br + sr - r;
    const u32 lval = bv + sv - v;
    u32 m = starts;
    while (m) {
        const u32 b = __ffs(m) - 1;
        m &= m - 1;
        s_rl[lrow] = (unsigned short)(threadIdx.x * 16 + b);
        s_rv[lrow] = (unsigned short)(lval + __popc(valid & ((1u << b) - 1)));
        lrow++;
    }
    __syncthreads();
    const RowTile T = tiles[rt];
    const u64 lvl0 = (u64)(rt - rtile_prefix[k]) * RT;
    // int32: a chunk holds fewer than 2^31 levels (checked when the plan is made)
    int32_t* rl = (int32_t*)C.row_level_off + T.row_base;
    int32_t* rv = (int32_t*)C.row_value_off + T.row_base;
    if (C.shares_rep) {  // the level offsets are the twin chunk's
        for (u32 j = threadIdx.x; j < tot_r; j += 256) *as_global(rv + j) = (int32_t)(T.val_base + s_rv[j]);
    } else {
        for (u32 j = threadIdx.x; j < tot_r; j += 256) {
            *as_global(rl + j) = (int32_t)(lvl0 + s_rl[j]);
            *as_global(rv + j) = (int32_t)(T.val_base + s_rv[j]);
        }
    }
}

__global__ __launch_bounds__(256) void k_rows_scan(const ChunkRec* __restrict__ chunks, const u32* __restrict__ rtile_prefix,
                                                   const u32* __restrict__ chunk_ids, u32 n_rchunks,
                                                   RowTile* __restrict__ tiles, ChunkDyn* __restrict__ chunk_dyn) {
    __shared__ u32 s_r[2][4], s_v[2][4];
    const u32 k = blockIdx.x;
    if (k >= n_rchunks) return;
    const u32 c = chunk_ids[k];
    const ChunkRec C = chunks[c];
    // 256 tiles of the chunk per round, one per thread, and the counts of NR rounds asked for together: a thread that went
    // through a slice of its own one tile after the other made 27 trips to memory, twice, for a chunk of 7000 tiles.
    // Per round a scan over the workgroup (four waves: a larger workgroup waits long for room on one CU while the long
    // kernels of the other streams run) on top of the totals of the rounds before.
    constexpr u32 NR = 16;
    const u32 t0 = rtile_prefix[k], t1 = rtile_prefix[k + 1];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    u64 run_r = 0, run_v = 0;
    u32 par = 0;  // the partial sums of consecutive rounds go to alternate slots: one barrier a round
    for (u32 g0 = t0; g0 < t1; g0 += 256 * NR) {
        u32 rr[NR], vv[NR];
#pragma unroll
        for (u32 q = 0; q < NR; q++) {
            const u32 t = g0 + q * 256 + tid;
            rr[q] = t < t1 ? tiles[t].rows : 0;  // at most RT each: a workgroup's sums fit 32 bits
            vv[q] = t < t1 ? tiles[t].vals : 0;
        }
#pragma unroll
        for (u32 q = 0; q < NR; q++) {
            if (g0 + q * 256 >= t1) break;  // the same for all threads
            const u32 t = g0 + q * 256 + tid;
            const u32 sr = sf_wave_scan(rr[q]), sv = sf_wave_scan(vv[q]);
            if (lane == 63) { s_r[par][wv] = sr; s_v[par][wv] = sv; }
            __syncthreads();
            u32 br = 0, bv = 0, tr = 0, tv = 0;
            for (u32 w = 0; w < 4; w++) {
                if (w < wv) { br += s_r[par][w]; bv += s_v[par][w]; }
                tr += s_r[par][w]; tv += s_v[par][w];
            }
            if (t < t1) {
                tiles[t].row_base = run_r + br + sr - rr[q];
                tiles[t].val_base = run_v + bv + sv - vv[q];
            }
            run_r += tr;
            run_v += tv;
            par ^= 1;
        }
    }
    if (tid == 0) {
        chunk_dyn[c].n_rows = run_r;
        if (!C.shares_rep) ((int32_t*)C.row_level_off)[run_r] = (int32_t)C.n_levels;
        ((int32_t*)C.row_value_off)[run_r] = (int32_t)run_v;
    }
}

// =====================================================================================
// launch wrappers (called from pqd_api.cpp)
// =====================================================================================
#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t launch_copy(hipStream_t st, const u8* src, const Seg* segs, const u32* blk_prefix, const u32* blk_grp, u32 n_segs,
                       u32 n_blocks, u32 max_grid) {
    if (n_blocks == 0) return hipSuccess;
    const u32 grid = n_blocks < max_grid ? n_blocks : max_grid;
    hipLaunchKernelGGL(k_copy, dim3(grid), dim3(CP_THREADS), 0, st, src, segs, blk_prefix, blk_grp, n_segs, n_blocks);
    LAUNCH_CHECK();
    return hipSuccess;
}

// Threads per tile of k_chain (levels, snappy elements).  Pass 1 is all sweeps and takes two waves per tile
// (levels: 60 -> 42 us on 4100 tiles); the marking pass has one lane chain over the sub-tiles, which twice as
// many sub-tiles make longer (73 -> 82 us), so it keeps one wave.
constexpr u32 CHAIN_NT = 128, CHAIN_NT_MARK = 64;
template <int KIND>
static hipError_t launch_chain_t(hipStream_t st, const u8* src, const ChainArgs& a, RunRec* run_tbl, u32* otile_first_run,
                                 const PageDyn* page_dyn, int* status, int phase) {
    constexpr u32 E = KIND == KIND_STRINGS ? CH_E_STRINGS : CH_E_LEVELS;
    if (a.n_regions == 0) return hipSuccess;
    // strings behind the single pass: only the chunks it gave up
    const u32* const gate = KIND == KIND_STRINGS ? a.str_failed : nullptr;
    const u32* const gate_any = KIND == KIND_STRINGS ? a.str_any : nullptr;
    if (phase == 0) {
        if (a.n_tiles) {
            if constexpr (KIND == KIND_STRINGS)
                hipLaunchKernelGGL(k_str_exits, dim3((a.n_tiles + STR_A * STR_ITER - 1) / (STR_A * STR_ITER)), dim3(CH_THREADS), 0, st, src, a.regions,
                                   a.tile_prefix, a.n_regions, a.n_tiles, a.exit_tbl, (u32x4*)a.str_cp, a.tile_region,
                                   (unsigned short*)a.str_index.cand,
                                   (unsigned short*)a.str_index.ncand, gate, gate_any);
            else
                hipLaunchKernelGGL((k_chain<KIND, false, CHAIN_NT>), dim3(a.n_tiles), dim3(CHAIN_NT), 0, st, src, a.regions,
                                   a.tile_prefix, a.n_regions, a.exit_tbl, a.tile_entry, run_tbl, otile_first_run, page_dyn,
                                   a.tile_region);
            LAUNCH_CHECK();
            hipLaunchKernelGGL((k_chain_groups<KIND>), dim3((u32)(((u64)a.n_groups * E + 63) / 64)), dim3(64), 0, st, src, a.regions,
                               a.group_prefix, a.n_regions, a.n_groups, a.exit_tbl, a.gexit, a.str_index, gate, gate_any);
            LAUNCH_CHECK();
        }
        const bool supers = a.n_supers > 0;
        if (supers) {
            hipLaunchKernelGGL((k_chain_supers<KIND>), dim3(a.n_supers * (E > 64 ? E / 64 : 1)), dim3(64), 0, st, src, a.regions,
                               a.super_prefix, a.n_regions, a.gexit, a.sexit, a.str_index, gate, gate_any);
            LAUNCH_CHECK();
        }
        // the region hop also runs for empty regions: it writes their (zero) totals
        hipLaunchKernelGGL((k_chain_resolve<KIND>), dim3((a.n_regions + 63) / 64), dim3(64), 0, st, src, a.regions, a.n_regions,
                           a.exit_tbl, a.gexit, a.super_prefix, supers ? a.sexit : nullptr, a.super_entry, a.group_entry,
                           a.totals, status, a.str_index, gate);
        LAUNCH_CHECK();
        if (supers) {
            hipLaunchKernelGGL((k_chain_super_groups<KIND>), dim3((a.n_supers + 63) / 64), dim3(64), 0, st, src, a.regions,
                               a.super_prefix, a.n_regions, a.n_supers, a.exit_tbl, a.gexit, a.super_entry, a.group_entry,
                               a.str_index, gate_any);
            LAUNCH_CHECK();
        }
        if (a.n_tiles) {
            hipLaunchKernelGGL((k_chain_tiles<KIND>), dim3((a.n_groups + 63) / 64), dim3(64), 0, st, src, a.regions,
                               a.group_prefix, a.n_regions, a.n_groups, a.exit_tbl, a.group_entry, a.tile_entry, a.str_index, gate_any);
            LAUNCH_CHECK();
        }
    } else if (a.n_tiles) {
        if constexpr (KIND == KIND_STRINGS)
            hipLaunchKernelGGL(k_str_mark, dim3(gate_any && (a.n_tiles + STR_MG - 1) / STR_MG > 8192 ? 8192 : (a.n_tiles + STR_MG - 1) / STR_MG), dim3(CH_THREADS), 0, st, src, a.regions,
                               a.tile_prefix, a.n_regions, a.n_tiles, a.tile_entry, (const u32x4*)a.str_cp, page_dyn, a.tile_region, status, gate_any);
        else
            hipLaunchKernelGGL((k_chain<KIND, true, CHAIN_NT_MARK>), dim3(a.n_tiles), dim3(CHAIN_NT_MARK), 0, st, src, a.regions, a.tile_prefix,
                               a.n_regions, a.exit_tbl, a.tile_entry, run_tbl, otile_first_run, page_dyn, a.tile_region);
        LAUNCH_CHECK();
    }
    return hipSuccess;
}

hipError_t launch_chain(hipStream_t st, int kind, const u8* src, const ChainArgs& a, RunRec* run_tbl, u32* otile_first_run,
                        const PageDyn* page_dyn, int* status, int phase) {
    if (kind == KIND_LEVELS) return launch_chain_t<KIND_LEVELS>(st, src, a, run_tbl, otile_first_run, page_dyn, status, phase);
    if (kind == KIND_SNAPPY) return launch_chain_t<KIND_SNAPPY>(st, src, a, run_tbl, otile_first_run, page_dyn, status, phase);
    return launch_chain_t<KIND_STRINGS>(st, src, a, run_tbl, otile_first_run, page_dyn, status, phase);
}

hipError_t launch_str_fused(hipStream_t st, const u8* src, const ChainArgs& a, PageDyn* str_pd, u8* ff_begin, size_t ff_bytes) {
    if (a.n_regions == 0) return hipSuccess;
    if (a.n_units) {
        hipLaunchKernelGGL(k_str_fused, dim3(a.n_units), dim3(SF_THREADS), 0, st, src, a.regions, a.units,
                           a.str_stat, a.str_failed, a.str_repairs, a.str_any, a.str_stamps);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_str_totals, dim3((a.n_regions + 63) / 64), dim3(64), 0, st, a.regions, a.n_regions, a.str_stat,
                       a.str_failed, a.totals, str_pd);
    LAUNCH_CHECK();
    if (ff_bytes) {
        hipLaunchKernelGGL(k_str_reset, dim3(1024), dim3(256), 0, st, (u32x4*)ff_begin, ff_bytes / 16, a.str_any);
        LAUNCH_CHECK();
    }
    return hipSuccess;
}

hipError_t launch_levels_expand(hipStream_t st, int esz, const u8* src, const Region* regions, const u32* otile_region,
                                const u32* otile_prefix, u32 n_regions, u32 n_otiles, const RunRec* run_tbl,
                                const u32* otile_first_run, const RegionTotal* totals, u32* otile_cnt, PageDyn* page_dyn) {
    if (n_otiles == 0) return hipSuccess;
    if (esz == 1) return hipErrorInvalidValue;  // one byte per value is k_levels (launch_rle_decode_u8)
    if (esz == 2)
        hipLaunchKernelGGL((k_levels_expand<2>), dim3(n_otiles), dim3(256), 0, st, src, regions, otile_prefix, n_regions, run_tbl,
                           otile_first_run, totals, otile_cnt);
    else
        hipLaunchKernelGGL((k_levels_expand<4>), dim3(n_otiles), dim3(256), 0, st, src, regions, otile_prefix, n_regions, run_tbl,
                           otile_first_run, totals, otile_cnt);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_page_scan(hipStream_t st, const ChunkRec* chunks, u32 n_chunks, const PageRec* pages, u32 n_pages,
                            PageDyn* page_dyn, ChunkDyn* chunk_dyn, int* status) {
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_page_scan, dim3(n_chunks), dim3(64), 0, st, chunks, n_chunks, pages, page_dyn, chunk_dyn, status);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_value_index(hipStream_t st, const ChunkRec* chunks, const PageRec* pages, const PageDyn* page_dyn,
                              const u32* page_ids, u32 n_pages, const u64* page_lvl_off) {
    if (n_pages == 0) return hipSuccess;
    hipLaunchKernelGGL(k_value_index, dim3(n_pages), dim3(256), 0, st, chunks, pages, page_dyn, page_ids, page_lvl_off);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_bools(hipStream_t st, const u8* src, const PageRec* pages, const u32* page_ids, const u32* btile_prefix,
                        const u32* btile_owner, u32 n_bpages, u32 n_btiles, const ChunkRec* chunks, const PageDyn* page_dyn,
                        const int* status) {
    if (n_btiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_bools, dim3(n_btiles), dim3(256), 0, st, src, pages, page_ids, btile_prefix, btile_owner, n_bpages, chunks, page_dyn,
                       status);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_scatter(hipStream_t st, int width, bool offsets, const u8* validity, const u8* value_index, const u8* values,
                          u8* dst, u64 n_slots, u64 n_words, u32 n_src) {
    if (n_slots == 0) return hipSuccess;
    const u32 per = 16 / (u32)width;
    const u64 n_thr = (n_slots + per - 1) / per, grid = (n_thr + 255) / 256;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    const u64* vb = (const u64*)validity;
    const u32* vx = (const u32*)value_index;
    if (offsets) hipLaunchKernelGGL((k_scatter<4, true>), dim3((u32)grid), dim3(256), 0, st, vb, vx, values, dst, n_slots, n_words, n_src);
    else if (width == 1) hipLaunchKernelGGL((k_scatter<1, false>), dim3((u32)grid), dim3(256), 0, st, vb, vx, values, dst, n_slots, n_words, n_src);
    else if (width == 4) hipLaunchKernelGGL((k_scatter<4, false>), dim3((u32)grid), dim3(256), 0, st, vb, vx, values, dst, n_slots, n_words, n_src);
    else if (width == 8) hipLaunchKernelGGL((k_scatter<8, false>), dim3((u32)grid), dim3(256), 0, st, vb, vx, values, dst, n_slots, n_words, n_src);
    else return hipErrorInvalidValue;
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_str_page_scan(hipStream_t st, const Region* regions, const u32* region_prefix, u32 n_str_chunks,
                                const RegionTotal* totals, PageDyn* str_pd, const u32* gate) {
    if (n_str_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_str_page_scan, dim3(n_str_chunks), dim3(64), 0, st, regions, region_prefix, n_str_chunks, totals, str_pd, gate);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_str_check(hipStream_t st, const PageRec* pages, const ChunkRec* chunks, const PageDyn* page_dyn,
                            const Region* regions, u32 n_regions, const RegionTotal* totals, int* status) {
    if (n_regions == 0) return hipSuccess;
    hipLaunchKernelGGL(k_str_check, dim3((n_regions + 255) / 256), dim3(256), 0, st, regions, n_regions, totals, pages, chunks,
                       page_dyn, status);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_rows(hipStream_t st, const ChunkRec* chunks, const u32* rtile_prefix, const u32* rtile_owner, const u32* chunk_ids,
                       u32 n_rchunks, u32 n_rtiles, RowTile* tiles, ChunkDyn* chunk_dyn) {
    if (n_rchunks == 0) return hipSuccess;
    if (n_rtiles) {
        hipLaunchKernelGGL(k_rows_count, dim3((n_rtiles + 3) / 4), dim3(256), 0, st, chunks, rtile_prefix, rtile_owner, chunk_ids, n_rchunks,
                           n_rtiles, tiles);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_rows_scan, dim3(n_rchunks), dim3(256), 0, st, chunks, rtile_prefix, chunk_ids, n_rchunks, tiles, chunk_dyn);
    LAUNCH_CHECK();
    if (n_rtiles) {
        hipLaunchKernelGGL((k_rows<true>), dim3(n_rtiles), dim3(256), 0, st, chunks, rtile_prefix, rtile_owner, chunk_ids, n_rchunks, tiles);
        LAUNCH_CHECK();
    }
    return hipSuccess;
}

hipError_t launch_snappy_tiles(hipStream_t st, const u8* src, const Region* regions, const u32* otile_region, u32 n_otiles,
                               const RunRec* run_tbl, const u32* otile_first_run, const RegionTotal* totals, int* status,
                               u32* tile_done) {
    if (n_otiles == 0) return hipSuccess;
    static const u32 strict = getenv("PQD_SNAPPY_STRICT") && atoi(getenv("PQD_SNAPPY_STRICT")) ? 1u : 0u;  // diagnostics, see the kernel
    // tile_done: n_otiles flags and the ticket counter behind them, all zero
    hipLaunchKernelGGL(k_snappy_ptr, dim3(n_otiles), dim3(SP_THREADS), 0, st, src, regions, otile_region, run_tbl, otile_first_run,
                       totals, status, tile_done, tile_done + n_otiles, n_otiles, strict);
    LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_level_block_sizes(hipStream_t st, const u8* src, const PageRec* pages, const ChunkRec* chunks, u32 n_pages,
                                    int* sizes) {
    if (n_pages == 0) return hipSuccess;
    hipLaunchKernelGGL(k_level_block_sizes, dim3((n_pages + 255) / 256), dim3(256), 0, st, src, pages, chunks, n_pages, sizes);
    LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace pqd
