// k_levels — single-pass decode of the hybrid RLE / bit-packed level blocks of the page path (gfx950).
//
// Reference behaviour reproduced: rle.Read with readLEB128 / readRLE / readRLEBitPacked
// (internal/rle/rle.go:159-290), bitpack.Unpack (internal/bitpack/bitpack.go:76-141), the truncation of the
// decoded levels to NumValues (fields.go:300,308) and valsFromDefs (fields.go:220-228).
//
// The position of a run header is only known after every header before it has been parsed.  A super-tile is
// LV_T bytes of one level block and belongs to one workgroup of LV_THREADS threads, 16 bytes per thread:
//   1. sweep       every byte is a possible header.  A thread sweeps its 16 bytes backwards and memoises, per
//                  start, where the chain that starts there leaves the 16 bytes (positions only, 16 bits each).
//   2. groups      the same per 256-byte group (16 sub-tiles), one round per sub-tile from the last to the first.
//   3. exits       for the first LV_TC starts of the super-tile: where the chain leaves the super-tile.
//   4. hand-off    the true chain enters where the super-tile before it says (one 8-byte granule per boundary,
//                  written and polled with agent-scope relaxed atomics).  The exit of the true entry is looked
//                  up and published at once, so the latency per super-tile on the chain of a block is one
//                  hand-off and one LDS lookup; everything below happens after the successor has been served.
//   5. resolve     one lane hops from group to group, one lane per group hops through its sub-tiles: every
//                  thread learns where the true chain enters its 16 bytes.
//   6. values      every thread walks its true runs and adds up their values; a workgroup scan and a second
//                  granule chain (values up to the end of each super-tile) turn that into output offsets.
//   7. expand      the true runs go to a list in LDS and the workgroup writes its part of the level array, 16
//                  values per thread and step (aligned 16-byte stores), counting def == max_def on the way.
// A start whose exit does not fit the 16-bit tables (a run that ends more than 64 KB behind the super-tile) is
// marked FAR; only the true chain ever resolves such a step, by parsing that header again with full arithmetic.
//
// Workgroups take their super-tile from a ticket counter, in an order in which every super-tile comes after the
// one before it in its block: a workgroup that waits for a granule waits for a workgroup that already runs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pqd_device.h"
#include "pqd_rle.h"

namespace pqd {

constexpr u32 LV_PITCH = LV_THREADS + 2;        // entries per memo row (row k = starts at offset k of every sub-tile)
constexpr u32 LV_MEMO = LV_SUB * LV_PITCH;      // entries per memo table
constexpr u32 LV_FAR = 0xFFFE, LV_ERR = 0xFFFF; // 16-bit exit codes: not representable / the chain fails here
constexpr u32 LV_GROUP = 256;                   // bytes per hop group
constexpr u32 LV_GSUB = LV_GROUP / LV_SUB;      // sub-tiles per group
constexpr u32 LV_NGROUPS = LV_T / LV_GROUP;
constexpr u32 LV_TC = 512;                      // tabulated entries of a super-tile: its first LV_TC bytes
constexpr u32 LV_MAXRUNS = LV_T / 2;            // a run that holds values takes at least 2 bytes (width >= 1)
constexpr u32 LV_WAVES = LV_THREADS / 64;
constexpr u32 LV_TAIL_MAX = 4096;               // bit-packed values past NumValues that are still examined
static_assert(LV_SUB == 16, "window loads, memo index and group rounds assume 16-byte sub-tiles");
static_assert(LV_GSUB * 4 == 64, "one wave = four groups in the group rounds");
static_assert(LV_TC <= LV_THREADS, "one thread per tabulated entry");
static_assert(2 * LV_MEMO * 2 >= (LV_MAXRUNS + 1) * 4 + 16 + LV_MAXRUNS * 4, "the run list reuses the memo tables");

static __device__ __forceinline__ u32 lv_idx(u32 rel) { return (rel & (LV_SUB - 1)) * LV_PITCH + (rel >> 4); }

// bytes k..k+3 of a 6-dword register window, k a compile-time constant after unrolling
static __device__ __forceinline__ u32 lv_win32(const u32 (&w)[6], int k) {
    const int d = k >> 2, sh = (k & 3) * 8;
    return sh ? (w[d] >> sh) | (w[d + 1] << (32 - sh)) : w[d];
}
// bytes k..k+7 and byte k+8 of the window for a run-time k in [0, 16)
static __device__ __forceinline__ void lv_win_dyn(const u32 (&w)[6], u32 k, u32& lo, u32& hi, u32& b8) {
    const u32 d = k >> 2, sh = (k & 3) * 8;
    const u32 a = d == 0 ? w[0] : d == 1 ? w[1] : d == 2 ? w[2] : w[3];
    const u32 b = d == 0 ? w[1] : d == 1 ? w[2] : d == 2 ? w[3] : w[4];
    const u32 c = d == 0 ? w[2] : d == 1 ? w[3] : d == 2 ? w[4] : w[5];
    lo = __builtin_amdgcn_alignbit(b, a, sh);
    hi = __builtin_amdgcn_alignbit(c, b, sh);
    b8 = (c >> sh) & 0xFF;
}

// One run of the true chain at position q, parsed from the thread's window (k = q - sub_lo): next header,
// values, and the record the expand step reads: an RLE run carries its value (width <= 8), a bit-packed run
// the position of its packed bytes.
static __device__ __forceinline__ void lv_run(const u32 (&w)[6], u32 k, u32 q, u32 L, u32 width, u32& nxt, u32& vals, u32& info) {
    u32 lo, hi, b8;
    lv_win_dyn(w, k, lo, hi, b8);
    rle_step_w32(lo, hi, q, L, width, nxt, vals, info);
    if (!(info & 1)) {
        const u32 d = (info >> 1) - q;  // the value byte sits 1..8 bytes after q
        u32 val = 0;
        if (width > 0 && nxt != ERR_POS) val = d < 4 ? (lo >> (8 * d)) & 0xFF : d < 8 ? (hi >> (8 * (d - 4))) & 0xFF : b8;
        info = val << 1;
    } else if (width == 0) info = 0;  // width 0 bit-packed groups are runs of zeros
}

// 0x80 in every byte of x that is zero
static __device__ __forceinline__ u64 lv_zero_bytes(u64 x) {
    return ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x) & 0x8080808080808080ull;
}

static __device__ __forceinline__ u64 lv_poll(const u64* g) {
    for (u32 spins = 0; spins < LV_SPIN_MAX; spins++) {
        const u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (x >> 32) return x;
        __builtin_amdgcn_s_sleep(2);
    }
    return 0;
}
static __device__ __forceinline__ void lv_publish(u64* g, u32 v) {
    __hip_atomic_store(g, (1ull << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(LV_THREADS) void k_levels(const u8* __restrict__ src_base, const LvArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short s_memo[2 * LV_MEMO];  // sweep memo | group memo; later the run list
    __shared__ unsigned short s_tc[LV_TC], s_entry[LV_THREADS], s_gentry[LV_NGROUPS];
    __shared__ u32 s_wn[LV_WAVES], s_wc[LV_WAVES];
    __shared__ u64 s_wv[LV_WAVES];
    __shared__ u32 s_ticket, s_e, s_ex, s_prefix, s_bad;
    unsigned short* const memo0 = s_memo;
    unsigned short* const la = s_memo + LV_MEMO;
    u32* const s_out = (u32*)s_memo;                                  // LV_MAXRUNS + 1 run starts
    u32* const s_info = (u32*)s_memo + ((LV_MAXRUNS + 1 + 3) & ~3u);  // LV_MAXRUNS run records

    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_ticket = atomicAdd(a.ticket, 1u);
    s_entry[tid] = 0xFFFF;
    if (tid < LV_NGROUPS) s_gentry[tid] = 0xFFFF;
    __syncthreads();
    const u32 ticket = s_ticket;
    if (ticket >= a.n_tiles) return;
    const LvTile ot = a.order[ticket];
    const Region R = a.regions[ot.region];
    const u32 L = R.len, width = R.width, j = ot.index;
    const u32 sb = j * LV_T, se = sb + LV_T;      // a level block is shorter than 2^31 bytes
    const u32 Lr = L - sb < LV_FAR ? L - sb : LV_FAR - 1;  // region end relative to the super-tile, as far as the tables reach
    const u8* const rsrc = src_base + R.src_off;
    const u32 n_out = (u32)R.n_out;
    const bool last = (u64)se >= L;               // the block ends in this super-tile

    // ---- my 16 bytes and 8 bytes of lookahead, in registers; bytes at or past L read as 0
    const u32 sub_lo = sb + tid * LV_SUB, sub_hi = sub_lo + LV_SUB;
    u32 w[6];
    if (sub_lo + 24 <= L) {
        const u32x4 x = *(const u32x4_u*)(rsrc + sub_lo);
        const u64 y = *(const u64_u*)(rsrc + sub_lo + 16);
        w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w; w[4] = (u32)y; w[5] = (u32)(y >> 32);
    } else {
#pragma unroll
        for (int d = 0; d < 6; d++) {
            const u32 p = sub_lo + 4 * d;
            u32 v = 0;
            if (p < L) {
                if (L - p >= 4) v = *(const u32_u*)(rsrc + p);
                else for (u32 k = 0; k < L - p; k++) v |= (u32)rsrc[p + k] << (8 * k);
            }
            w[d] = v;
        }
    }

    // ---- 1. sweep: where the chain of every start leaves my 16 bytes
#pragma unroll
    for (int k = (int)LV_SUB - 1; k >= 0; k--) {
        const u32 p = sub_lo + k;
        u32 nxt, vals, info;
        rle_step_w32(lv_win32(w, k), lv_win32(w, k + 4), p, L, width, nxt, vals, info);
        const bool valid = p < L;
        const bool in = valid & (nxt < sub_hi) & (nxt < L);  // lands on a later start of mine: its answer is final
        const u32 mi = (u32)k * LV_PITCH + tid;
        const u32 mj = in ? (nxt - sub_lo) * LV_PITCH + tid : mi;
        const u32 t2 = memo0[mj];
        const u32 d = nxt - sb;
        u32 enc = nxt == ERR_POS ? LV_ERR : d >= LV_FAR ? LV_FAR : d;
        if (!valid) enc = LV_ERR;
        memo0[mi] = (unsigned short)(in ? t2 : enc);
    }
    // ---- 2. groups: a wave owns 1 KiB = 4 groups; round s settles the starts of sub-tile s of every group
    // (lane = group * 16 + start).  Only this wave touches these entries: no workgroup barrier between rounds.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int s = (int)LV_GSUB - 1; s >= 0; s--) {
        const u32 g = lane >> 4, k = lane & 15;
        const u32 rel = wv * 1024 + g * LV_GROUP + (u32)s * LV_SUB + k;
        const u32 gend = wv * 1024 + (g + 1) * LV_GROUP;
        const u32 x = memo0[lv_idx(rel)];
        const bool hop = (x < gend) & (x < Lr);
        const u32 y = la[lv_idx(hop ? x : rel)];
        la[lv_idx(rel)] = (unsigned short)(hop ? y : x);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- 3. exits of the super-tile for its first LV_TC starts
    if (tid < LV_TC) {
        u32 p = tid;
        while (p < LV_T && p < Lr) p = la[lv_idx(p)];
        s_tc[tid] = (unsigned short)p;
    }
    __syncthreads();

    // one step of the true chain with full arithmetic, from the bytes in memory
    auto step_precise = [&](u32 p) -> u32 {
        const u64 x = load_u64_guarded(rsrc, p, L);
        u32 nxt, v, info;
        rle_step_w32((u32)x, (u32)(x >> 32), p, L, width, nxt, v, info);
        return nxt;
    };
    // the true chain is at p (sb <= p < min(se, L)): where it leaves p's sub-tile / p's group
    auto hop_sub = [&](u32 p) -> u32 {
        const u32 x = memo0[lv_idx(p - sb)];
        if (x < LV_FAR) return sb + x;
        if (x == LV_ERR) return ERR_POS;
        const u32 send = (p | (LV_SUB - 1)) + 1;  // sb is a multiple of LV_SUB
        while (p < send && p < L) p = step_precise(p);  // ERR_POS ends the loop too
        return p;
    };
    auto hop_group = [&](u32 p) -> u32 {
        const u32 x = la[lv_idx(p - sb)];
        if (x < LV_FAR) return sb + x;
        if (x == LV_ERR) return ERR_POS;
        const u32 gend = (p | (LV_GROUP - 1)) + 1;
        while (p < gend && p < L) p = hop_sub(p);
        return p;
    };

    // ---- 4. hand-off: entry from the super-tile before, exit to the one after; 5a. group entries
    if (tid == 0) {
        u32 e = 0;
        bool timeout = false;
        if (j > 0) {
            const u64 x = lv_poll(a.pos_g + R.lv_base + j - 1);
            timeout = !(x >> 32);
            e = timeout ? ERR_POS : (u32)x;
        }
        u32 ex;
        if (e == ERR_POS || e >= se || e >= L) ex = e;  // failed before / nothing starts here
        else {
            const u32 r = e - sb;
            const u32 x = r < LV_TC ? s_tc[r] : LV_FAR;
            if (x < LV_FAR) ex = sb + x;
            else if (x == LV_ERR) ex = ERR_POS;
            else {
                u32 p = e;
                while (p < se && p < L) p = hop_group(p);
                ex = p;
            }
        }
        lv_publish(a.pos_g + R.lv_base + j, ex);
        if (timeout) set_status(a.status, ST_INTERNAL, ot.region);
        else if (ex == ERR_POS && e != ERR_POS) set_status(a.status, ST_CORRUPT, ot.region);
        s_e = e;
        s_ex = ex;
        if (ex != ERR_POS && e < se && e < L) {
            u32 p = e;
            while (p < se && p < L) {
                s_gentry[(p - sb) / LV_GROUP] = (unsigned short)(p - sb);
                p = hop_group(p);
            }
        }
    }
    __syncthreads();
    const u32 e = s_e, ex = s_ex;
    if (ex == ERR_POS) return;  // the chain failed in this block: status is set, nothing more is written
    const bool active = e < se && e < L;  // the true chain starts at least one run here

    // ---- 5b. sub-tile entries: one lane per group
    if (active && tid < LV_NGROUPS) {
        const u32 r = s_gentry[tid];
        if (r != 0xFFFF) {
            u32 p = sb + r;
            const u32 gend = sb + (tid + 1) * LV_GROUP;
            while (p < gend && p < L) {
                s_entry[(p - sb) / LV_SUB] = (unsigned short)(p - sb);
                p = hop_sub(p);
            }
        }
    }
    __syncthreads();

    // ---- 6. values: runs that hold values and the values of all runs that start in my 16 bytes
    const u32 my_entry = active ? s_entry[tid] : 0xFFFF;
    u32 n_runs = 0;
    u64 n_vals = 0;
    if (my_entry != 0xFFFF) {
        u32 q = sb + my_entry;
        while (q < sub_hi && q < L) {
            u32 nxt, vals, info;
            lv_run(w, q - sub_lo, q, L, width, nxt, vals, info);
            n_runs += vals != 0;
            n_vals += vals;
            q = nxt;  // never ERR_POS: the exit of the super-tile is known to be good
        }
    }
    u32 sn = n_runs;
    u64 sv = n_vals;
    for (int d = 1; d < 64; d <<= 1) {
        const u32 tn = __shfl_up(sn, d);
        const u64 tv = __shfl_up(sv, d);
        if (lane >= (u32)d) { sn += tn; sv += tv; }
    }
    if (lane == 63) { s_wn[wv] = sn; s_wv[wv] = sv; }
    __syncthreads();
    u32 base_n = 0, tot_n = 0;
    u64 base_v = 0, tot_v = 0;
#pragma unroll
    for (u32 q = 0; q < LV_WAVES; q++) {
        if (q < wv) { base_n += s_wn[q]; base_v += s_wv[q]; }
        tot_n += s_wn[q];
        tot_v += s_wv[q];
    }
    if (tid == 0) {
        u32 prefix = 0;
        bool timeout = false;
        if (j > 0) {
            const u64 x = lv_poll(a.val_g + R.lv_base + j - 1);
            timeout = !(x >> 32);
            prefix = (u32)x;
        }
        const u64 incl = (u64)prefix + tot_v;
        lv_publish(a.val_g + R.lv_base + j, incl < 0xFFFFFFFFull ? (u32)incl : 0xFFFFFFFFu);
        if (timeout) set_status(a.status, ST_INTERNAL, ot.region);
        else if (last && incl < n_out) set_status(a.status, ST_CORRUPT, ot.region);  // defs[:NumValues] out of range
        s_prefix = prefix;
        s_bad = 0;
    }
    __syncthreads();
    if (!active) return;
    const u32 prefix = s_prefix;
    const u64 end64 = (u64)prefix + tot_v;
    const u32 o_lo = prefix < n_out ? prefix : n_out, o_hi = end64 < n_out ? (u32)end64 : n_out;

    // ---- 7a. the run list (the memo tables are dead: their last readers ran before the barriers above)
    u32 m;
    if (width == 0) {  // every value is 0: one run covers the super-tile's part
        m = 1;
        if (tid == 0) { s_out[0] = o_lo; s_info[0] = 0; s_out[1] = o_hi; }
    } else {
        m = tot_n;
        if (my_entry != 0xFFFF) {
            u32 q = sb + my_entry, rank = base_n + sn - n_runs;
            u64 o = (u64)prefix + base_v + sv - n_vals;
            bool bad = false;
            while (q < sub_hi && q < L) {
                u32 nxt, vals, info;
                lv_run(w, q - sub_lo, q, L, width, nxt, vals, info);
                if (vals) {
                    s_out[rank] = o < n_out ? (u32)o : n_out;
                    s_info[rank] = info;
                    rank++;
                    // Values past NumValues are dropped (fields.go:300,308), but valsFromDefs counts them
                    // (fields.go:311): a page whose surplus holds max_def is refused instead of counted differently.
                    if (R.count_level != ERR_POS && o + vals > n_out) {
                        const u64 t_lo = o > n_out ? o : n_out, t_n = o + vals - t_lo;
                        if (!(info & 1)) bad = bad || (info >> 1) == R.count_level;
                        else if (t_n > LV_TAIL_MAX) bad = true;
                        else {
                            const u64 cl = (u64)(R.count_level & 0xFF) * 0x0101010101010101ull;
                            for (u32 i = 0; i < (u32)t_n; i += 16) {
                                const u32x4 v = run16(rsrc, L, width, (u32)o, info, (u32)t_lo + i);
                                const u32 nb = (u32)t_n - i < 16 ? (u32)t_n - i : 16;  // bytes of v that are surplus values
                                const u64 k_lo = nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1;
                                const u64 k_hi = nb >= 16 ? ~0ull : nb > 8 ? (1ull << (8 * (nb - 8))) - 1 : 0;
                                const u64 x_lo = (v.x | ((u64)v.y << 32)) ^ cl, x_hi = (v.z | ((u64)v.w << 32)) ^ cl;
                                bad = bad || (lv_zero_bytes(x_lo) & k_lo) || (lv_zero_bytes(x_hi) & k_hi);
                            }
                        }
                    }
                }
                o += vals;
                q = nxt;
            }
            if (bad) s_bad = 1;
        }
        if (tid == 0) s_out[m] = o_hi;
    }
    __syncthreads();
    if (tid == 0 && s_bad) set_status(a.status, ST_NONCANONICAL, ot.region);
    if (o_hi <= o_lo) return;

    // ---- 7b. expand [o_lo, o_hi): blocks of 16 values at 16-byte aligned addresses of the level array
    const u32 mis = (u32)((uintptr_t)(R.out + o_lo) & 15);
    const u32 vbase = o_lo - mis;                 // level index of block 0's first byte (wraps below 0: only differences are used)
    u8* const dbase = R.out + o_lo - mis;         // its address, 16-byte aligned
    const u32 v_end = mis + (o_hi - o_lo);
    const u32 n_blk = (v_end + 15) >> 4;
    const bool counting = R.count_level != ERR_POS;
    const u64 cl64 = (u64)(R.count_level & 0xFF) * 0x0101010101010101ull;
    u32 cnt = 0;
    for (u32 b = tid; b < n_blk; b += LV_THREADS) {
        const u32 v0 = b * 16;
        const u32 lo_v = v0 > mis ? v0 : mis, hi_v = v0 + 16 < v_end ? v0 + 16 : v_end;
        const u32 x0 = vbase + v0, lo = vbase + lo_v, hi = vbase + hi_v;  // level indices
        // last run that starts at or before lo (run 0 starts at o_lo <= lo)
        u32 ra = 0, rb = m;
        while (rb - ra > 1) {
            const u32 mid = (ra + rb) >> 1;
            if (s_out[mid] <= lo) ra = mid; else rb = mid;
        }
        u64 acc_lo = 0, acc_hi = 0;
        u32 r = ra;
        do {
            const u32 so = s_out[r], sn_ = s_out[r + 1];
            const u32 start = so > lo ? so : lo;
            if (sn_ > start) merge16(acc_lo, acc_hi, run16(rsrc, L, width, so, s_info[r], start), start - x0);
            r++;
        } while (r < m && s_out[r] < hi);
        u8* const dst = dbase + v0;
        if (hi_v - lo_v == 16) {
            u32x4 v;
            v.x = (u32)acc_lo; v.y = (u32)(acc_lo >> 32); v.z = (u32)acc_hi; v.w = (u32)(acc_hi >> 32);
            *(u32x4*)dst = v;
        } else {
            for (u32 k = lo_v - v0; k < hi_v - v0; k++) dst[k] = (u8)(k < 8 ? acc_lo >> (8 * k) : acc_hi >> (8 * (k - 8)));
        }
        if (counting) {
            const u32 k0 = lo_v - v0, k1 = hi_v - v0;  // bytes [k0, k1) of the block are mine
            auto keep = [](u32 a0, u32 a1) -> u64 {     // 0xFF in bytes [a0, a1) of a 64-bit word, 0 <= a0 <= a1 <= 8
                const u64 hi_m = a1 >= 8 ? ~0ull : (1ull << (8 * a1)) - 1;
                const u64 lo_m = a0 >= 8 ? ~0ull : (1ull << (8 * a0)) - 1;
                return hi_m & ~lo_m;
            };
            const u64 keep_lo = keep(k0 < 8 ? k0 : 8, k1 < 8 ? k1 : 8);
            const u64 keep_hi = keep(k0 > 8 ? k0 - 8 : 0, k1 > 8 ? k1 - 8 : 0);
            cnt += __popcll(lv_zero_bytes(acc_lo ^ cl64) & keep_lo) + __popcll(lv_zero_bytes(acc_hi ^ cl64) & keep_hi);
        }
    }
    if (counting) {
        for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d);
        if (lane == 0) s_wc[wv] = cnt;
        __syncthreads();
        if (tid == 0) {
            u32 c = 0;
#pragma unroll
            for (u32 q = 0; q < LV_WAVES; q++) c += s_wc[q];
            if (c && a.page_dyn) atomicAdd(&a.page_dyn[R.page].nn, c);
        }
    }
}

hipError_t launch_levels(hipStream_t st, const u8* src, const LvArgs& a) {
    if (a.n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_levels, dim3(a.n_tiles), dim3(LV_THREADS), 0, st, src, a);
    return hipGetLastError();
}

}  // namespace pqd
