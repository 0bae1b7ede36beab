// k_levels — single-pass decode of the hybrid RLE / bit-packed level blocks of the page path (gfx950).
//
// Reference behaviour reproduced: rle.Read with readLEB128 / readRLE / readRLEBitPacked
// (internal/rle/rle.go:159-290), bitpack.Unpack (internal/bitpack/bitpack.go:76-141), the truncation of the
// decoded levels to NumValues (fields.go:300,308) and valsFromDefs (fields.go:220-228).
//
// The position of a run header is only known after every header before it has been parsed.  A super-tile is
// LV_T bytes of one level block and belongs to one workgroup of LV_THREADS threads, 16 bytes per thread:
//   1. sweep       every byte is a possible header.  A thread sweeps its 16 bytes backwards and memoises, per
//                  start, where the chain that starts there leaves the 16 bytes and how many values it passes
//                  (16 bits each).
//   2. groups      the same per 256-byte group (16 sub-tiles), one round per sub-tile from the last to the first.
//   3. exits       for the first LV_TC starts of the super-tile: where the chain leaves the super-tile.
//   4. hand-off    the true chain enters where the super-tile before it says (one 8-byte granule per boundary,
//                  written and polled with agent-scope relaxed atomics).  The exit of the true entry is looked
//                  up and published at once, so the latency per super-tile on the chain of a block is one
//                  hand-off and one LDS lookup; everything below happens after the successor has been served.
//   5. resolve     one lane hops from group to group, one lane per group hops through its sub-tiles: every
//                  thread learns where the true chain enters its 16 bytes.
//   6. values      the memo of a thread's true entry says how many values its runs hold; a workgroup scan and a
//                  second granule chain (values up to the end of each super-tile) turn that into output offsets.
//   7. expand      window by window (65536 bits of output): every thread parses its true runs and puts their
//                  bits into a dense stream of `width` bits per value in LDS (short runs through a register
//                  accumulator, long ones word by word, very long ones with the whole wave); the stream is then
//                  unpacked to one byte per value, 16 values per thread and step, with aligned 16-byte stores,
//                  counting def == max_def and writing the validity bitmap on the way.
// A start whose exit does not fit the 16-bit tables (a run that ends more than 64 KB behind the super-tile) is
// marked FAR; only the true chain ever resolves such a step, by parsing that header again with full arithmetic.
//
// Workgroup b takes super-tile b of a launch order in which every super-tile comes after the one before it in its
// block: a workgroup that waits for a granule waits for a workgroup with a lower index, which was started before it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>

#include "pqd_device.h"
#include "pqd_rle.h"

#ifndef LV_NT_STORE
#define LV_NT_STORE 1
#endif
#ifndef LV_DIRECT_OR
#define LV_DIRECT_OR 1
#endif
namespace pqd {
#if PQD_LV_TILE_BYTES != 8192
namespace lv4k {  // the second build of the kernel keeps its names apart
#endif

constexpr u32 LV_PITCH = LV_THREADS + 1;        // entries per memo row (row k = starts at offset k of every sub-tile)
constexpr u32 LV_MEMO = LV_SUB * LV_PITCH;      // entries per memo table
constexpr u32 LV_FAR = 0xFFFE, LV_ERR = 0xFFFF; // 16-bit exit codes: not representable / the chain fails here
constexpr u32 LV_VSAT = 0xFFFF;                 // 16-bit value count: this many or more, or unknown
constexpr u32 LV_GROUP = 256;                   // bytes per hop group
constexpr u32 LV_GSUB = LV_GROUP / LV_SUB;      // sub-tiles per group
constexpr u32 LV_NGROUPS = LV_T / LV_GROUP;
constexpr u32 LV_TC = 128;  // tabulated entries of a super-tile: its first LV_TC bytes (a bit-packed run of 63 groups of width 2 reaches 127 bytes into the next super-tile)
constexpr u32 LV_DENSE_BITS = 65536;            // bits of the dense level stream per expand window
constexpr u32 LV_DENSE_BIAS = 128;              // the stream starts at this bit: a block may begin up to 15 values before the window
constexpr u32 LV_DENSE_WORDS = LV_DENSE_BIAS / 32 + LV_DENSE_BITS / 32 + 6;
constexpr u32 LV_SEG_SERIAL = 32;               // a run that covers more stream words than this is written by its whole wave
constexpr u32 LV_WAVES = LV_THREADS / 64;
constexpr u32 LV_TAIL_MAX = 4096;               // bit-packed values past NumValues that are still examined
static_assert(LV_SUB == 16, "window loads, memo index and group rounds assume 16-byte sub-tiles");
static_assert(LV_GSUB * 4 == 64, "one wave = four groups in the group rounds");
static_assert(LV_TC <= LV_THREADS, "one thread per tabulated entry");
static_assert(LV_MEMO * 4 >= ((LV_DENSE_WORDS + 3) & ~3u) * 4 + LV_T + 16 + LV_THREADS * 4, "the dense stream, the byte image and the anchors reuse the sweep memo");

static __device__ __forceinline__ u32 lv_idx(u32 rel) { return (rel & (LV_SUB - 1)) * LV_PITCH + (rel >> 4); }

// bytes k..k+3 of a 6-dword register window, k a compile-time constant after unrolling
static __device__ __forceinline__ u32 lv_win32(const u32 (&w)[6], int k) {
    const int d = k >> 2, sh = (k & 3) * 8;
    return sh ? (w[d] >> sh) | (w[d + 1] << (32 - sh)) : w[d];
}
// bytes r..r+7 and byte r+8 of the super-tile's LDS image (r < LV_T).  The runs of the true chain are parsed from
// there: picking registers of the thread's window by a run-time index would put the window into scratch memory.
static __device__ __forceinline__ void lv_win_img(const u32* img, u32 r, u32& lo, u32& hi, u32& b8) {
    const u32 d = r >> 2, sh = (r & 3) * 8;
    const u32 a = img[d], b = img[d + 1], c = img[d + 2];
    lo = __builtin_amdgcn_alignbit(b, a, sh);
    hi = __builtin_amdgcn_alignbit(c, b, sh);
    b8 = (c >> sh) & 0xFF;
}

// A run header of at most 4 bytes (fewer than 2^27 values or groups) at position p, from the 4 bytes at p: the
// same answers as rle_step_w32.  n = header bytes (0: the header is longer, nothing else is valid), nxt = next
// header or ERR_POS, vals = values of the run, bp = bit-packed.
// TAIL = false: the caller knows that p + 8 <= L, so neither the header nor the value byte of an RLE run nor the first
// packed byte can lie behind the block end, and the checks for that are left out.
template <bool TAIL = true>
static __device__ __forceinline__ void lv_header4(u32 lo, u32 p, u32 L, u32 w, u32& n, u32& nxt, u32& vals, bool& bp) {
    const u32 s = ~lo & 0x80808080u;                                  // the stop bits
    const u32 nb = ((u32)__builtin_ctz(s | 0x80000000u) >> 3) + 1;    // header bytes
    const u32 a = lo & (s ^ (s - 1));                                 // the header's bytes
    const u32 h = (a & 0x7F) | ((a >> 1) & 0x3F80) | ((a >> 2) & 0x1FC000) | ((a >> 3) & 0xFE00000);
    const u32 q = p + nb, cnt = h >> 1, bw = (w + 7) >> 3;
    bp = (h & 1) != 0;
    const u32 prod = w * cnt;                                         // < 2^30
    const u32 bp_nxt = prod >= L - q ? L : q + prod;
    bool bad = false;
    if (TAIL) {
        bad = nb > L - p;                                             // header cut off by the block end
        bad = bad | (bp ? ((w != 0) & (q >= L)) : (bw > L - q));      // io.EOF on the packed bytes / on the value
    }
    n = s ? nb : 0;
    nxt = bad ? ERR_POS : bp ? bp_nxt : q + bw;
    vals = bad ? 0 : bp ? cnt * 8 : cnt;
}

// The sweep's header parse: a header of one or two bytes (fewer than 2^13 values or groups: every bit-packed run of
// the reference writer, rle.go:69-71, and RLE runs of up to 8191 values) at position p that lies at least 8 bytes in
// front of the block end, from the 2 bytes at p.  n = header bytes, 0 for a longer header (nothing else is valid then: the
// sweep marks the start FAR, and the true chain, should it come here, parses it with full arithmetic).  Half the
// instructions of lv_header4, and the sweep runs it on every byte.
static __device__ __forceinline__ void lv_header2(u32 lo, u32 p, u32 L, u32 w, u32 bw, u32& n, u32& nxt, u32& vals) {
    const bool one = !(lo & 0x80), two = !(lo & 0x8000);
    const u32 h = one ? lo & 0x7F : (lo & 0x7F) | ((lo >> 1) & 0x3F80);
    n = one ? 1 : two ? 2 : 0;
    const u32 q = p + n, cnt = h >> 1;
    const bool bp = (h & 1) != 0;
    const u32 bp_nxt = q + __umul24(w, cnt);  // < 2^16 bytes on
    nxt = bp ? (bp_nxt < L ? bp_nxt : L) : q + bw;
    vals = bp ? cnt * 8 : cnt;  // < 2^16
}

// One run of the true chain at position q, parsed from the image (r = q - start of the super-tile): next header, values,
// and how to produce them: info = value << 1 for an RLE run (width <= 8), position of the packed bytes << 1 | 1
// for a bit-packed run, whose first 32 packed bits are pay32 when pay_ok.
static __device__ __forceinline__ void lv_run(const u32* img, u32 r, u32 q, u32 L, u32 width, u32& nxt, u32& vals, u32& info,
                                              u32& pay32, bool& pay_ok, u64* pay64 = nullptr, u32* pay_bits = nullptr) {
    u32 lo, hi, b8, n;
    bool bp;
    lv_win_img(img, r, lo, hi, b8);
    lv_header4(lo, q, L, width, n, nxt, vals, bp);
    if (n) {
        const u64 after64 = (lo | ((u64)hi << 32)) >> (8 * n);  // the 8 - n bytes behind the header
        const u32 after = (u32)after64;
        if (pay64) { *pay64 = after64; *pay_bits = 64 - 8 * n; }
        pay32 = after;
        pay_ok = true;
        info = bp ? (width ? ((q + n) << 1) | 1u : 0u) : (width ? (after & 0xFF) << 1 : 0u);  // width 0: runs of zeros
    } else {
        rle_step_w32(lo, hi, q, L, width, nxt, vals, info);
        pay32 = 0;
        pay_ok = false;
        if (pay64) { *pay64 = 0; *pay_bits = 0; }
        if (!(info & 1)) {
            const u32 d = (info >> 1) - q;  // the value byte sits 5..8 bytes after q
            u32 val = 0;
            if (width > 0 && nxt != ERR_POS) val = d < 4 ? (lo >> (8 * d)) & 0xFF : d < 8 ? (hi >> (8 * (d - 4))) & 0xFF : b8;
            info = val << 1;
        } else if (width == 0) info = 0;
    }
}

// 16 consecutive values (one per result byte) of the dense stream dn (W bits per value), starting at bit `bit`
template <int W>
static __device__ __forceinline__ u32x4 lv_dense16_w(const u32* dn, u32 bit) {
    constexpr int NX = (16 * W + 31) / 32;
    const u32 d = bit >> 5, sh = bit & 31;
    u32 raw[NX + 1], x[NX + 1];
#pragma unroll
    for (int k = 0; k <= NX; k++) raw[k] = dn[d + k];
#pragma unroll
    for (int k = 0; k < NX; k++) x[k] = __builtin_amdgcn_alignbit(raw[k + 1], raw[k], sh);
    x[NX] = 0;
    u32 o[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int b = k * W, q = b >> 5, r = b & 31;
        u32 v;
        if (r + W <= 32) v = (x[q] >> r) & ((1u << W) - 1);
        else v = __builtin_amdgcn_alignbit(x[q + 1], x[q], r) & ((1u << W) - 1);
        o[k >> 2] |= v << (8 * (k & 3));
    }
    return u32x4{o[0], o[1], o[2], o[3]};
}
static __device__ __forceinline__ u32x4 lv_dense16(const u32* dn, u32 bit, u32 w) {
    switch (w) {
    case 0: return u32x4{0, 0, 0, 0};
    case 1: return lv_dense16_w<1>(dn, bit);
    case 2: return lv_dense16_w<2>(dn, bit);
    case 3: return lv_dense16_w<3>(dn, bit);
    case 4: return lv_dense16_w<4>(dn, bit);
    case 5: return lv_dense16_w<5>(dn, bit);
    case 6: return lv_dense16_w<6>(dn, bit);
    case 7: return lv_dense16_w<7>(dn, bit);
    default: return lv_dense16_w<8>(dn, bit);
    }
}

// Word jw of the dense stream for the part [bb0, bb1) (stream bits) of a run: an RLE run repeats its value
// (pt = the value in every `width` bits, rr = phase of the word's first bit inside a value), a bit-packed run
// copies its packed bytes (stream bit bb0 = packed bit sb0 of the bytes at py), from the LDS image of the
// super-tile where they lie inside it, else from memory; packed bytes past the block end read as 0
// (rle.go:203-213).
static __device__ __forceinline__ u32 lv_word_bits(u32 jw, u32 bb0, u32 bb1, bool is_bp, u64 pt, u32 rr, u32 py, u64 sb0,
                                                   const u8* rsrc, u32 L, const u32* img, u32 sb, u32 se) {
    u32 x;
    if (is_bp) {
        const int64_t S = (int64_t)sb0 + (int64_t)(jw * 32) - (int64_t)bb0;  // packed bit that lands on bit 0 of word jw
        const u64 sbit = S >= 0 ? (u64)S : 0;  // S < 0: packed bit 0 lands on bit -S of the word (what lies below bb0 is masked off)
        const u64 byte = (u64)py + (sbit >> 3);
        if (byte <= (u64)se) {  // inside the image (py >= sb): two of its words hold the 39 bits needed
            const u32 rb = (u32)byte - sb, sh = (rb & 3) * 8 + ((u32)sbit & 7);
            x = __builtin_amdgcn_alignbit(img[(rb >> 2) + 1], img[rb >> 2], sh);
        } else x = (u32)(load_u64_guarded(rsrc, byte < 0xFFFFFFFFull ? (u32)byte : 0xFFFFFFFFu, L) >> (sbit & 7));
        if (S < 0) x <<= (u32)(-S);
    } else x = (u32)(pt >> rr);
    u32 m = 0xFFFFFFFFu;
    if (jw == bb0 >> 5) m &= 0xFFFFFFFFu << (bb0 & 31);
    if (jw == (bb1 - 1) >> 5) m &= 0xFFFFFFFFu >> (31 - ((bb1 - 1) & 31));
    return x & m;
}

// bit i of the result = bit 8 i + 7 of x
static __device__ __forceinline__ u32 lv_msb_gather(u64 x) { return (u32)(((x >> 7) * 0x0102040810204080ull) >> 56); }
// 0x80 in every byte of x that is zero
static __device__ __forceinline__ u64 lv_zero_bytes(u64 x) {
    return ~(((x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | x) & 0x8080808080808080ull;
}

// waits for a granule; bounded by time (profilers and other tenants stretch a wait, not its polls): 0 = gave up
static __device__ __forceinline__ u64 lv_poll(const u64* g) {
    const u64 t0 = wall_clock64();
    for (;;) {
        const u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (x >> 32) return x;
        if (wall_clock64() - t0 > LV_WAIT_TICKS) return 0;
        __builtin_amdgcn_s_sleep(2);
    }
}
// a wait that ran out: status = {ST_INTERNAL, region, super-tile waited for, 0 position granule / 1 value granule}
static __device__ __forceinline__ void lv_timeout(int* status, u32 region, u32 tile, int which) {
    if (atomicCAS(status, 0, ST_INTERNAL) == 0) { status[1] = (int)region; status[2] = (int)tile; status[3] = which; }
}
static __device__ __forceinline__ void lv_publish(u64* g, u32 v) {
    __hip_atomic_store(g, (1ull << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define LV_STAMP(i) do { if (a.stamps && tid == 0) a.stamps[(u64)ticket * LV_STAMPS + (i)] = wall_clock64(); } while (0)

__global__ __launch_bounds__(LV_THREADS) void k_levels(const u8* __restrict__ src_base, const LvArgs a) {
    // sweep memo: exit | values << 16 per start; later the dense stream and the byte image of the super-tile
    __shared__ __attribute__((aligned(16))) u32 s_memo[LV_MEMO];
    __shared__ unsigned short s_la[LV_MEMO];  // group memo: exit per start
    __shared__ unsigned short s_tc[LV_TC], s_entry[LV_THREADS], s_gentry[LV_NGROUPS];
    __shared__ u32 s_wc[LV_WAVES];
    __shared__ u64 s_wv[LV_WAVES];
    __shared__ u32 s_e, s_ex, s_prefix, s_bad, s_fill, s_anch;
    u32* const memo0 = s_memo;
    unsigned short* const la = s_la;

    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    s_entry[tid] = 0xFFFF;
    if (tid < LV_NGROUPS) s_gentry[tid] = 0xFFFF;  // both are next touched behind several barriers
    // Workgroup b takes entry b of the launch order.  The dispatcher starts workgroups in the order of their indices (on
    // every XCD in the order of the indices it is given), so the workgroup with the lowest index that has not finished
    // is running or is the next to start, and it waits for nobody: a wait for a lower index always ends.  A shared ticket
    // counter gave the same order at the price of one atomic per workgroup (2 us of every workgroup's life, and about
    // 12 ns each for all workgroups of the launch one after the other).
    const u32 ticket = blockIdx.x;
    if (ticket >= a.n_tiles) return;
    const LvTile ot = a.order[ticket];
    if (ot.region == ERR_POS) return;  // a position of the order that holds no super-tile
    LV_STAMP(0);
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

    LV_STAMP(1);
    // ---- 1. sweep: where the chain of every start leaves my 16 bytes, and the values it passes on the way
    auto sweep = [&](auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;  // false: my 16 bytes and 8 more lie inside the block
        if constexpr (!TAIL) {
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            const u32 bw = (width + 7) >> 3;
#pragma unroll
            for (int k = (int)LV_SUB - 1; k >= 0; k--) {
                u32 n, nxt, vals;
                lv_header2(lv_win32(w, k), sub_lo + k, L, width, bw, n, nxt, vals);
                const bool in = (n != 0) & (nxt < sub_hi);  // lands on a later start of mine: its answer is final
                const u32 mi = (u32)k * LV_PITCH + tid;
                const u32 t2 = *(const u32*)((const u8*)memo0 + (in ? __umul24(nxt - sub_lo, LV_PITCH * 4) + tid * 4 : mi * 4));
                const u32 d = nxt - sb;
                const u32 enc = n == 0 ? LV_FAR : d < LV_FAR ? d : LV_FAR;  // a longer header: parsed with full arithmetic if the true chain comes here
                // exit in the low half, values in the high half, added with saturation (LV_VSAT: that many or more, or unknown)
                const u32 base = in ? t2 : enc, add = (in | (enc < LV_FAR)) ? vals << 16 : LV_VSAT << 16;
                const u16x2 sum = __builtin_elementwise_add_sat(__builtin_bit_cast(u16x2, base), __builtin_bit_cast(u16x2, add));
                memo0[mi] = __builtin_bit_cast(u32, sum);
            }
            return;
        }
#pragma unroll
        for (int k = (int)LV_SUB - 1; k >= 0; k--) {
            const u32 p = sub_lo + k;
            u32 n, nxt, vals;
            bool bp;
            lv_header4<TAIL>(lv_win32(w, k), p, L, width, n, nxt, vals, bp);
            const bool valid = !TAIL || p < L;
            const bool in = valid & (n != 0) & (nxt < sub_hi) & (!TAIL || nxt < L);  // lands on a later start of mine: its answer is final
            const u32 mi = (u32)k * LV_PITCH + tid;
            const u32 mj = in ? (nxt - sub_lo) * LV_PITCH + tid : mi;
            const u32 t2 = memo0[mj];
            const u32 d = nxt - sb;
            u32 enc = TAIL && nxt == ERR_POS ? LV_ERR : d >= LV_FAR ? LV_FAR : d;
            if (n == 0) enc = LV_FAR;  // a longer header: parsed with full arithmetic if the true chain comes here
            if (!valid) enc = LV_ERR;
            u32 v = enc >= LV_FAR ? LV_VSAT : vals;
            if (in) v += t2 >> 16;
            v = v < LV_VSAT ? v : LV_VSAT;
            memo0[mi] = (in ? t2 & 0xFFFF : enc) | (v << 16);
        }
    };
    if (sub_lo + 24 <= L) sweep(std::false_type{});
    else sweep(std::true_type{});
    LV_STAMP(2);
    // ---- 2. groups: a wave owns 1 KiB = 4 groups; round s settles the starts of sub-tile s of every group
    // (lane = group * 16 + start).  Only this wave touches these entries: no workgroup barrier between rounds.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int s = (int)LV_GSUB - 1; s >= 0; s--) {
        const u32 g = lane >> 4, k = lane & 15;
        const u32 rel = wv * 1024 + g * LV_GROUP + (u32)s * LV_SUB + k;
        const u32 gend = wv * 1024 + (g + 1) * LV_GROUP;
        const u32 x = memo0[lv_idx(rel)] & 0xFFFF;
        const bool hop = (x < gend) & (x < Lr);
        const u32 y = la[lv_idx(hop ? x : rel)];
        la[lv_idx(rel)] = (unsigned short)(hop ? y : x);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    LV_STAMP(3);
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
        const u32 x = memo0[lv_idx(p - sb)] & 0xFFFF;
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

    LV_STAMP(4);
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
        LV_STAMP(5);
        lv_publish(a.pos_g + R.lv_base + j, ex);
        LV_STAMP(6);
        if (timeout) lv_timeout(a.status, ot.region, j - 1, 0);
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

    LV_STAMP(7);
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

    LV_STAMP(8);
    // ---- 6. values of the true runs that start in my 16 bytes: the memo of my entry knows them, unless they
    // are 65535 or more or a header needs full arithmetic; then the runs are parsed for it
    const u32 my_entry = active ? s_entry[tid] : 0xFFFF;
    u64 n_vals = my_entry != 0xFFFF ? memo0[lv_idx(my_entry)] >> 16 : 0;
    __syncthreads();  // the sweep memo is dead from here on
    // the super-tile's bytes go to LDS: the true runs are parsed from there
    u32* const dense = s_memo;                                  // LV_DENSE_WORDS
    u32* const img = dense + ((LV_DENSE_WORDS + 3) & ~3u);      // LV_T / 4 + 2 words: the super-tile and 8 bytes behind it
    u32* const anch = img + LV_T / 4 + 4;                       // LV_THREADS words, see 7a
    *(u32x4*)(img + tid * 4) = u32x4{w[0], w[1], w[2], w[3]};
    if (tid == LV_THREADS - 1) { img[LV_T / 4] = w[4]; img[LV_T / 4 + 1] = w[5]; }
    __syncthreads();
    if (n_vals == LV_VSAT) {
        n_vals = 0;
        u32 q = sb + my_entry;
        while (q < sub_hi && q < L) {
            u32 nxt, vals, info, pay32;
            bool pay_ok;
            lv_run(img, q - sb, q, L, width, nxt, vals, info, pay32, pay_ok);
            n_vals += vals;
            q = nxt;  // never ERR_POS: the exit of the super-tile is known to be good
        }
    }
    u64 sv = n_vals;
    for (int d = 1; d < 64; d <<= 1) {
        const u64 tv = __shfl_up(sv, d);
        if (lane >= (u32)d) sv += tv;
    }
    if (lane == 63) s_wv[wv] = sv;
    __syncthreads();
    u64 base_v = 0, tot_v = 0;
#pragma unroll
    for (u32 q = 0; q < LV_WAVES; q++) {
        if (q < wv) base_v += s_wv[q];
        tot_v += s_wv[q];
    }
    LV_STAMP(9);
    if (tid == 0) {
        u32 prefix = 0;
        bool timeout = false;
        if (j > 0) {
            const u64 x = lv_poll(a.val_g + R.lv_base + j - 1);
            timeout = !(x >> 32);
            prefix = (u32)x;
        }
        LV_STAMP(10);
        const u64 incl = (u64)prefix + tot_v;
        lv_publish(a.val_g + R.lv_base + j, incl < 0xFFFFFFFFull ? (u32)incl : 0xFFFFFFFFu);
        if (timeout) lv_timeout(a.status, ot.region, j - 1, 1);
        else if (last && incl < n_out && !(R.otile_base & LV_ALLOW_SHORT)) set_status(a.status, ST_CORRUPT, ot.region);  // defs[:NumValues] out of range
        s_prefix = prefix;
        s_bad = 0;
        s_fill = 0;
    }
    __syncthreads();
    if (!active) return;
    const u32 prefix = s_prefix;
    const u64 my_first = base_v + sv - n_vals;  // local index (within the super-tile's values) of my first run's first value
    const bool counting = R.count_level != ERR_POS;
    const bool bitmap_only = (R.otile_base & LV_BITMAP_ONLY) != 0;  // repetition levels: which levels are 0, for the row offsets
    LV_STAMP(11);

    // Values past NumValues are dropped (fields.go:300,308), but valsFromDefs counts them (fields.go:311): a
    // page whose surplus holds max_def is refused instead of being counted differently.
    if (counting && !bitmap_only && (u64)prefix + tot_v > n_out) {
        bool bad = false;
        if (my_entry != 0xFFFF) {
            u64 o = (u64)prefix + my_first;
            u32 q = sb + my_entry;
            while (q < sub_hi && q < L) {
                u32 nxt, vals, info, pay32;
                bool pay_ok;
                lv_run(img, q - sb, q, L, width, nxt, vals, info, pay32, pay_ok);
                if (vals && width && o + vals > n_out) {
                    const u64 t_lo = o > n_out ? o : n_out, t_n = o + vals - t_lo;
                    if (!(info & 1)) bad = bad || (info >> 1) == R.count_level;
                    else if (t_n > LV_TAIL_MAX) bad = true;
                    else {
                        const u64 cl = (u64)(R.count_level & 0xFF) * 0x0101010101010101ull;
                        for (u32 k = 0; k < (u32)t_n; k += 16) {
                            const u32x4 v = run16(rsrc, L, width, (u32)o, info, (u32)t_lo + k);
                            const u32 nb = (u32)t_n - k < 16 ? (u32)t_n - k : 16;  // bytes of v that are surplus values
                            const u64 k_lo = nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1;
                            const u64 k_hi = nb >= 16 ? ~0ull : nb > 8 ? (1ull << (8 * (nb - 8))) - 1 : 0;
                            const u64 x_lo = (v.x | ((u64)v.y << 32)) ^ cl, x_hi = (v.z | ((u64)v.w << 32)) ^ cl;
                            bad = bad || (lv_zero_bytes(x_lo) & k_lo) || (lv_zero_bytes(x_hi) & k_hi);
                        }
                    }
                }
                o += vals;
                q = nxt;
            }
        }
        if (bad) s_bad = 1;
        __syncthreads();
        if (tid == 0 && s_bad) set_status(a.status, ST_NONCANONICAL, ot.region);
    }

    // ---- 7. expand.  U values of this super-tile are wanted: level indices [prefix, prefix + U).  They are
    // produced window by window: the runs are first assembled into a dense stream of `width` bits per value in
    // LDS, then the stream is unpacked to one byte per value.  Bit LV_DENSE_BIAS of the stream is the window's
    // first value.
    const u32 U = prefix >= n_out ? 0 : (tot_v < (u64)(n_out - prefix) ? (u32)tot_v : n_out - prefix);
    const u32 V = width ? (LV_DENSE_BITS / width) : LV_DENSE_BITS;  // values per window
    const u64 cl64 = (u64)(R.count_level & 0xFF) * 0x0101010101010101ull;
    u8* const vbits = counting ? R.out2 : nullptr;  // validity bitmap of the chunk, bit i = level i of the chunk
    const u64 rep_mul = (width & (width - 1)) || !width ? 0 : ~0ull / ((1ull << width) - 1);  // 1 in every width-th bit
    const bool odd_w = (width & (width - 1)) != 0;
    u32 cnt = 0;
    for (u32 W0 = 0; W0 < U; W0 += V) {
        const u32 W1 = U - W0 < V ? U : W0 + V;
        const u32 n_words = LV_DENSE_BIAS / 32 + ((W1 - W0) * width + 31) / 32 + 6;  // a few spare words: blocks read past the end
        for (u32 i = tid; i < n_words; i += LV_THREADS) dense[i] = 0;
        anch[tid] = 0;
        if (tid == 0) s_anch = 0;
        __syncthreads();
        // ---- 7a. assemble: every thread parses its runs; the loop has the same trip count for all lanes of a
        // wave because a very long run is written by all of them
        if (width) {
            u32 q = my_entry != 0xFFFF ? sb + my_entry : ERR_POS;
            u64 ra = my_first;
            // consecutive short runs collect in a register: bits [0, accn) of acc belong at word accw of the stream
            u64 acc = 0;
            u32 accn = 0, accw = 0;
            auto acc_flush = [&]() {
                if ((u32)acc) atomicOr(&dense[accw], (u32)acc);
                if (accn > 32 && (u32)(acc >> 32)) atomicOr(&dense[accw + 1], (u32)(acc >> 32));
                acc = 0;
                accn = 0;
            };
            auto acc_seek = [&](u32 bit) {  // the next bits appended go to stream bit `bit`
                if (accn == 0 || accw * 32 + accn != bit) {
                    acc_flush();
                    accw = bit >> 5;
                    accn = bit & 31;
                }
            };
            auto append = [&](u32 bits, u32 n) {
                acc |= (u64)bits << accn;
                accn += n;
                if (accn >= 32) {
                    if ((u32)acc) atomicOr(&dense[accw], (u32)acc);
                    acc >>= 32;
                    accw++;
                    accn -= 32;
                }
            };
            // A run that is not of the usual kind (below) is put off until the thread has gone through its sub-tile, so
            // that the lanes of a wave take the long way together: `slow` writes one such run of every lane that has one.
            auto slow = [&](bool on, u32 qs, u64 ras) {
                u32 b0 = 0, b1 = 0, pay = 0;
                u64 pat = 0, sbit0 = 0;
                bool bp = false, is_long = false;
                if (on) {
                    u32 nxt, vals, info, pay32, pay_bits;
                    u64 pay64;
                    bool pay_ok;
                    lv_run(img, qs - sb, qs, L, width, nxt, vals, info, pay32, pay_ok, &pay64, &pay_bits);
                    const u64 rb = ras + vals;
                    if (vals && ras < W1 && rb > W0) {
                        const u32 s = ras > W0 ? (u32)ras : W0, t = rb < W1 ? (u32)rb : W1;
                        b0 = LV_DENSE_BIAS + (s - W0) * width;
                        b1 = LV_DENSE_BIAS + (t - W0) * width;
                        const u32 nbits = b1 - b0;
                        bp = info & 1;
                        bool live = true;
                        if (!bp) {
                            const u32 v = info >> 1;
                            if (rep_mul) pat = v * rep_mul;
                            else { pat = v; for (u32 sft = width; sft < 64; sft <<= 1) pat |= pat << sft; }
                            if (s == W0 && t == W1) s_fill = v + 1;  // the whole window is this run: stage 7b skips the stream
                            live = v != 0;  // zeros are there already
                        } else {
                            pay = info >> 1;
                            sbit0 = (u64)(s - ras) * width;
                        }
                        u32 bits_left = nbits;  // what this thread still has to put into the stream itself
                        if (bp && s == ras && t == rb && nbits > 64) {
                            // A bit-packed run that lies in the window as a whole is its packed bytes, copied.  This thread
                            // copies the ones in its own 16 bytes; the threads of the sub-tiles behind it copy theirs (7a',
                            // told where through `anch`: stream bit | bytes << 17 | first byte << 22); what lies behind the
                            // super-tile is left to this thread again.
                            const u32 plen = nbits >> 3, pr = pay - sb, own_end = (tid + 1) * LV_SUB;
                            const u32 own = pr >= own_end ? 0 : (own_end - pr < plen ? own_end - pr : plen);
                            u32 done = own, pos = pr + own, obit = b0 + 8 * own;
                            while (done < plen && pos < LV_T) {
                                const u32 sub = pos / LV_SUB, skip = pos % LV_SUB;
                                const u32 nb = LV_SUB - skip < plen - done ? LV_SUB - skip : plen - done;
                                anch[sub] = obit | (nb << 17) | (skip << 22);
                                s_anch = 1;
                                obit += 8 * nb; pos += nb; done += nb;
                            }
                            if (own) {
                                acc_seek(b0);
                                u64 sbit = 0;
                                for (u32 rem = 8 * own; rem;) {
                                    const u32 take = rem < 32 ? rem : 32;
                                    const u32 rbt = pr + (u32)(sbit >> 3), sh = (rbt & 3) * 8;  // inside the image
                                    append(__builtin_amdgcn_alignbit(img[(rbt >> 2) + 1], img[rbt >> 2], sh) & (0xFFFFFFFFu >> (32 - take)), take);
                                    sbit += take;
                                    rem -= take;
                                }
                            }
                            b0 += 8 * done;
                            sbit0 += 8 * (u64)done;
                            bits_left = nbits - 8 * done;
                            live = bits_left != 0;
                        }
                        is_long = live && (((b0 & 31) + bits_left + 31) >> 5) > LV_SEG_SERIAL;
                        if (live && !is_long) {
                            // the run's bits join the accumulator, at most 32 at a time
                            acc_seek(b0);
                            if (!bp) {
                                u32 ph = 0;
                                for (u32 rem = bits_left; rem;) {
                                    const u32 take = rem < 32 ? rem : 32;
                                    append((u32)(pat >> ph) & (0xFFFFFFFFu >> (32 - take)), take);
                                    rem -= take;
                                    if (odd_w) { ph += 32 % width; if (ph >= width) ph -= width; }
                                }
                            } else if (bits_left <= 32 && pay_ok && s == ras && bits_left == nbits) {
                                append(pay32 & (0xFFFFFFFFu >> (32 - bits_left)), bits_left);  // the packed bits are in my registers
                            } else {
                                u64 sbit = sbit0;
                                for (u32 rem = bits_left; rem;) {
                                    const u32 take = rem < 32 ? rem : 32;
                                    const u64 byte = (u64)pay + (sbit >> 3);
                                    u32 x;
                                    if (byte <= (u64)se) {  // inside the image: two of its words hold the 39 bits needed
                                        const u32 rbt = (u32)byte - sb, sh = (rbt & 3) * 8 + ((u32)sbit & 7);
                                        x = __builtin_amdgcn_alignbit(img[(rbt >> 2) + 1], img[rbt >> 2], sh);
                                    } else x = (u32)(load_u64_guarded(rsrc, byte < 0xFFFFFFFFull ? (u32)byte : 0xFFFFFFFFu, L) >> (sbit & 7));
                                    append(x & (0xFFFFFFFFu >> (32 - take)), take);
                                    sbit += take;
                                    rem -= take;
                                }
                            }
                        }
                    }
                }
                u64 lm = __ballot(is_long);
                while (lm) {
                    const int l = __ffsll((unsigned long long)lm) - 1;
                    lm &= lm - 1;
                    const u32 c_b0 = __shfl(b0, l), c_b1 = __shfl(b1, l), c_pay = __shfl(pay, l);
                    const u64 c_sb = __shfl(sbit0, l), c_pat = __shfl(pat, l);
                    const bool c_bp = __shfl((u32)bp, l) != 0;
                    const u32 c_j0 = c_b0 >> 5, c_j1 = (c_b1 - 1) >> 5;
                    for (u32 jw = c_j0 + lane; jw <= c_j1; jw += 64) {
                        const u32 rr = odd_w ? ((jw - LV_DENSE_BIAS / 32) * 32) % width : 0;
                        const u32 x = lv_word_bits(jw, c_b0, c_b1, c_bp, c_pat, rr, c_pay, c_sb, rsrc, L, img, sb, se);
                        if (x) atomicOr(&dense[jw], x);
                    }
                }
            };
            u32 dq0 = ERR_POS, dq1 = ERR_POS;  // the runs put off
            u64 dra0 = 0, dra1 = 0;
            for (;;) {
                const bool act = q < sub_hi && q < L;
                if (!__any(act)) break;
                u32 nxt = ERR_POS;
                bool now = false;  // a third run put off: written at once
                const u64 ra_run = ra;
                if (act) {
                    u32 vals, info, pay32, pay_bits;
                    u64 pay64;
                    bool pay_ok;
                    lv_run(img, q - sb, q, L, width, nxt, vals, info, pay32, pay_ok, &pay64, &pay_bits);
                    const u64 rb = ra + vals;
                    // the usual run: short, in the window as a whole, its bits at hand (an RLE value, or packed bytes
                    // that came with the header): two appends at most, no loops
                    const u32 nb_all = vals <= 64 ? __umul24(vals, width) : 0xFFFFFFFFu;
                    const bool quick = vals && ra >= W0 && rb <= W1 && nb_all <= ((info & 1) ? pay_bits : 64u) &&
                                       !(ra == W0 && rb == W1);
                    if (quick) {
                        const u32 qb0 = LV_DENSE_BIAS + __umul24((u32)ra - W0, width);
                        u64 bits;
                        if (info & 1) bits = pay64;
                        else {
                            const u32 v = info >> 1;
                            if (rep_mul) { const u32 p32 = v * (u32)rep_mul; bits = p32 | ((u64)p32 << 32); }  // width 1, 2, 4 or 8
                            else { bits = v; for (u32 sft = width; sft < 64; sft <<= 1) bits |= bits << sft; }
                        }
                        if (nb_all < 64) bits &= (1ull << nb_all) - 1;
                        if (bits) {  // zeros are there already
#if LV_DIRECT_OR
                            // straight into the stream: up to three words, each with one LDS atomic
                            const u32 j0 = qb0 >> 5, sh = qb0 & 31;
                            const u64 lo64 = bits << sh;
                            const u32 top = sh ? (u32)(bits >> (64 - sh)) : 0;
                            if ((u32)lo64) atomicOr(&dense[j0], (u32)lo64);
                            if ((u32)(lo64 >> 32)) atomicOr(&dense[j0 + 1], (u32)(lo64 >> 32));
                            if (top) atomicOr(&dense[j0 + 2], top);
#else
                            acc_seek(qb0);
                            const u32 n1 = nb_all < 32 ? nb_all : 32;
                            append((u32)bits, n1);
                            if (nb_all > 32) append((u32)(bits >> 32), nb_all - 32);
#endif
                        }
                    } else if (vals && ra < W1 && rb > W0) {
                        if (dq0 == ERR_POS) { dq0 = q; dra0 = ra; }
                        else if (dq1 == ERR_POS) { dq1 = q; dra1 = ra; }
                        else now = true;
                    }
                    ra = rb;
                }
                if (__any(now)) slow(now, q, ra_run);
                if (act) q = nxt;
            }
            LV_STAMP(14);
            if (__any(dq0 != ERR_POS)) slow(dq0 != ERR_POS, dq0, dra0);
            if (__any(dq1 != ERR_POS)) slow(dq1 != ERR_POS, dq1, dra1);
            acc_flush();
            LV_STAMP(15);
            // ---- 7a'. packed bytes of runs that started in an earlier sub-tile: every thread copies its own
            __syncthreads();
            const u32 an = s_anch ? anch[tid] : 0;  // s_anch is the same for all threads: no anchors, no second barrier
            if (an) {
                const u32 obit = an & 0x1FFFF, nb = (an >> 17) & 31, r = tid * LV_SUB + (an >> 22);
                const u32 d = r >> 2, sh = (r & 3) * 8, j0 = obit >> 5, s2 = obit & 31;
                u32 raw[5], x[4];
#pragma unroll
                for (int k = 0; k < 5; k++) raw[k] = img[d + k];  // at most 2 words behind the image: masked off below
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const u32 have = nb > 4u * k ? nb - 4u * k : 0;  // bytes of word k that belong to the run
                    const u32 v = __builtin_amdgcn_alignbit(raw[k + 1], raw[k], sh);
                    x[k] = have >= 4 ? v : have ? v & ((1u << (8 * have)) - 1) : 0;
                }
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const u32 lo = k ? x[k - 1] : 0, hi = k < 4 ? x[k] : 0;
                    const u32 o = s2 ? __builtin_amdgcn_alignbit(hi, lo, 32 - s2) : hi;
                    if (o) atomicOr(&dense[j0 + k], o);
                }
            }
            if (s_anch) __syncthreads();
        } else __syncthreads();
        LV_STAMP(12);
        // ---- 7b. unpack [W0, W1) to level indices [g_lo, g_hi): blocks of 16 values at 16-byte aligned addresses
        const u32 g_lo = prefix + W0, g_hi = prefix + W1;
        const u32 fill = s_fill;
        const u32 mis = (u32)((uintptr_t)(R.out + g_lo) & 15);
        u8* const dbase = R.out + g_lo - mis;
        const u32 v_end = mis + (g_hi - g_lo);
        const u32 n_blk = (v_end + 15) >> 4;
        for (u32 b = tid; b < n_blk; b += LV_THREADS) {
            const u32 v0 = b * 16;
            const u32 lo_v = v0 > mis ? v0 : mis, hi_v = v0 + 16 < v_end ? v0 + 16 : v_end;
            u32x4 v;
            u32 eq = 0x10000;  // bit k: value k of the block equals count_level; 0x10000: not known yet
            const u32 cl = R.count_level & 0xFF;
            if (fill) {
                const u32 f4 = (fill - 1) * 0x01010101u;
                v = u32x4{f4, f4, f4, f4};
                eq = fill - 1 == cl ? 0xFFFFu : 0u;
            } else if (width <= 2 && width) {
                // the usual widths: the comparison with count_level is made on the packed bits, not on the bytes
                const u32 bit = (u32)((int)LV_DENSE_BIAS + ((int)v0 - (int)mis) * (int)width);
                const u32 x = __builtin_amdgcn_alignbit(dense[(bit >> 5) + 1], dense[bit >> 5], bit & 31);  // 16 values
                if (width == 1) {
                    // 4 bits -> 4 bytes: n * 0x204081 puts bit i of n at bit 8 i (and others elsewhere)
                    u32 o[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) o[i] = __umul24((x >> (4 * i)) & 15, 0x204081u) & 0x01010101u;
                    v = u32x4{o[0], o[1], o[2], o[3]};
                    eq = cl == 1 ? x & 0xFFFFu : cl == 0 ? ~x & 0xFFFFu : 0u;
                } else {
                    v = lv_dense16_w<2>(dense, bit);
                    u32 y = 0;
                    if (cl < 4) {
                        const u32 z = x ^ (cl * 0x55555555u);       // a value equals cl where both of its bits are 0
                        y = ~(z | (z >> 1)) & 0x55555555u;          // bit 2 k: value k equals cl
                        y = (y | (y >> 1)) & 0x33333333u;           // the even bits, squeezed together
                        y = (y | (y >> 2)) & 0x0F0F0F0Fu;
                        y = (y | (y >> 4)) & 0x00FF00FFu;
                        y = (y | (y >> 8)) & 0xFFFFu;
                    }
                    eq = y;
                }
            } else v = lv_dense16(dense, (u32)((int)LV_DENSE_BIAS + ((int)v0 - (int)mis) * (int)width), width);
            u8* const dst = dbase + v0;
            const u64 acc_lo = v.x | ((u64)v.y << 32), acc_hi = v.z | ((u64)v.w << 32);
#if LV_NT_STORE
            if (hi_v - lo_v == 16) __builtin_nontemporal_store(v, as_global((u32x4*)dst));
#else
            if (hi_v - lo_v == 16) *as_global((u32x4*)dst) = v;
#endif
            else for (u32 k = lo_v - v0; k < hi_v - v0; k++) *as_global(dst + k) = (u8)(k < 8 ? acc_lo >> (8 * k) : acc_hi >> (8 * (k - 8)));
            if (counting) {
                // bit k: value k of the block equals max_def (fields.go:220-228); bits [k0, k1) are mine
                const u32 k0 = lo_v - v0, k1 = hi_v - v0;
                if (eq == 0x10000) eq = lv_msb_gather(lv_zero_bytes(acc_lo ^ cl64)) | (lv_msb_gather(lv_zero_bytes(acc_hi ^ cl64)) << 8);
                const u32 mine = eq & ((0xFFFFu >> (16 - k1)) & (0xFFFFu << k0));
                cnt += __popc(mine);
                if (vbits) {
                    const u64 li = R.run_base + g_lo + v0 - mis;  // chunk level index of the block's first value: a multiple of 16
                    if (k1 - k0 == 16) *as_global((unsigned short*)vbits + (li >> 4)) = (unsigned short)mine;
                    else if (mine) __hip_atomic_fetch_or(as_global((u32*)vbits + (li >> 5)), mine << (li & 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // a block shared with another workgroup or page
                }
            }
        }
        if (W1 < U) {
            __syncthreads();  // the stream is rewritten
            if (tid == 0) s_fill = 0;
        }
    }
    LV_STAMP(13);
    if (counting && !bitmap_only) {
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

#if PQD_LV_TILE_BYTES != 8192
}  // namespace lv4k
hipError_t launch_levels_4k(hipStream_t st, const u8* src, const LvArgs& a) {
    if (a.n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(lv4k::k_levels, dim3(a.n_tiles), dim3(LV_THREADS), 0, st, src, a);
    return hipGetLastError();
}
}  // namespace pqd
#else
hipError_t launch_levels(hipStream_t st, const u8* src, const LvArgs& a, u32 tile_bytes) {
    if (a.n_tiles == 0) return hipSuccess;
    if (tile_bytes == LV_T_LARGE_BATCH) return launch_levels_4k(st, src, a);
    if (tile_bytes != LV_T) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_levels, dim3(a.n_tiles), dim3(LV_THREADS), 0, st, src, a);
    return hipGetLastError();
}

// =====================================================================================
// One level block decoded on its own (rle.Read, internal/rle/rle.go:159-195, widths 0..8, one byte per value):
// k_levels over a single region whose tables live in a scratch buffer:
//   [LvsHead][order: n_tiles LvTile][pos_g: n_tiles u64][val_g: n_tiles u64]
// k_lv_setup writes the tables and zeroes the granules, k_lv_finish reports {values the block holds (saturated
// at 2^32 - 1), status} as two int64.  The scratch buffer is either allocated by the caller (pqd_rle_decode) or
// the static array below (pqd_rle_decode_noalloc), which every call on a device shares.
// =====================================================================================
struct LvsHead {
    Region region;
    u32 ticket;
    u32 pad[5];
    int status[4];
};
static_assert(sizeof(LvsHead) % 16 == 0, "the tables behind the head stay aligned");
static __host__ __device__ size_t lvs_bytes(u32 n_tiles) { return sizeof(LvsHead) + (size_t)n_tiles * (sizeof(LvTile) + 16); }
static __host__ __device__ u32 lvs_tiles(size_t len) { return len ? (u32)((len + LV_T - 1) / LV_T) : 1; }

constexpr u32 LVS_MAX_TILES = (u32)(PQD_NOALLOC_MAX_BLOCK / LV_T);
constexpr size_t LVS_STATIC_BYTES = sizeof(LvsHead) + (size_t)LVS_MAX_TILES * (sizeof(LvTile) + 16);
// Scratch of the entry point that allocates nothing.  In host code this name is only a handle for
// hipGetSymbolAddress: the device address is looked up once per device (lv_scratch_init) and used from then on.
static __device__ __align__(256) unsigned char scratch[LVS_STATIC_BYTES];
constexpr int LVS_MAX_DEVICES = 64;
// Per device: the array's address, and who used it last.  Contexts are created from any host thread (one per reader), and
// every context of a device shares the array: the slot is filled under its mutex, and a call that comes on another stream
// than the call before it is queued behind that call's event, so overlapping calls are serialised on the device
// instead of racing on the tables.
hipError_t launch_rle_decode_u8(hipStream_t st, void* sc_, int width, const u8* runs, u32 len, u8* out, u32 max_values,
                                long long* result_device);
struct LvScratchSlot {
    std::mutex mu;
    void* ptr = nullptr;
    hipEvent_t done = nullptr;     // recorded behind the last call's kernels
    hipStream_t last = nullptr;    // the stream of that call
    bool used = false;
};
static LvScratchSlot g_scratch[LVS_MAX_DEVICES];

hipError_t lv_scratch_init(int device) {  // the device is current
    if (device < 0 || device >= LVS_MAX_DEVICES) return hipErrorInvalidDevice;
    LvScratchSlot& S = g_scratch[device];
    std::lock_guard<std::mutex> lock(S.mu);
    if (S.ptr) return hipSuccess;
    void* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    e = hipEventCreateWithFlags(&S.done, hipEventDisableTiming);
    if (e != hipSuccess) return e;
    S.ptr = ptr;
    return hipSuccess;
}
void* lv_scratch_ptr(int device) {
    if (device < 0 || device >= LVS_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch[device].mu);
    return g_scratch[device].ptr;
}
// One decode over the device's static scratch, queued on `st` (no allocation: the event exists since lv_scratch_init).
hipError_t launch_rle_decode_u8_static(int device, hipStream_t st, int width, const u8* runs, u32 len, u8* out, u32 max_values,
                                       long long* result_device) {
    if (device < 0 || device >= LVS_MAX_DEVICES) return hipErrorInvalidDevice;
    LvScratchSlot& S = g_scratch[device];
    std::lock_guard<std::mutex> lock(S.mu);
    if (!S.ptr) return hipErrorNotInitialized;
    hipError_t e = hipSuccess;
    if (S.used && S.last != st) e = hipStreamWaitEvent(st, S.done, 0);  // behind the call before, whoever made it
    if (e == hipSuccess) e = launch_rle_decode_u8(st, S.ptr, width, runs, len, out, max_values, result_device);
    if (e == hipSuccess) e = hipEventRecord(S.done, st);
    if (e == hipSuccess) { S.last = st; S.used = true; }
    return e;
}
size_t lv_scratch_bytes(size_t runs_len) { return lvs_bytes(lvs_tiles(runs_len)); }

__global__ void k_lv_setup(unsigned char* sc, u32 n_tiles, u32 width, u32 len, u8* out, u32 max_values) {
    LvsHead* const H = (LvsHead*)sc;
    LvTile* const order = (LvTile*)(sc + sizeof(LvsHead));
    u64* const pos_g = (u64*)(order + n_tiles);
    u64* const val_g = pos_g + n_tiles;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_tiles) {
        order[i] = LvTile{0, i};
        pos_g[i] = 0;
        val_g[i] = 0;
    }
    if (i == 0) {
        Region R;
        R.src_off = 0; R.len = len; R.width = width; R.tile_base = 0;
        R.otile_base = LV_ALLOW_SHORT;  // fewer values than max_values is no error here
        R.run_base = 0; R.n_out = max_values; R.out = out; R.out2 = nullptr; R.page = 0;
        R.count_level = ERR_POS; R.group_base = 0; R.lv_base = 0;
        H->region = R;
        H->ticket = 0;
        H->status[0] = H->status[1] = H->status[2] = H->status[3] = 0;
    }
}
__global__ void k_lv_finish(const unsigned char* sc, u32 n_tiles, long long* result) {
    const LvsHead* const H = (const LvsHead*)sc;
    const u64* const val_g = (const u64*)(sc + sizeof(LvsHead) + (size_t)n_tiles * sizeof(LvTile)) + n_tiles;
    const int st = H->status[0];
    result[0] = st ? 0 : (long long)(u32)val_g[n_tiles - 1];
    result[1] = st == 0 ? 0 : st == ST_CORRUPT ? -6 : st == ST_NONCANONICAL ? -7 : st == ST_INTERNAL ? -2 : -5;  // pqd_status
}

// Queues the decode of one block on `st`; sc = scratch of at least lv_scratch_bytes(len) bytes, 16-byte aligned.
hipError_t launch_rle_decode_u8(hipStream_t st, void* sc_, int width, const u8* runs, u32 len, u8* out, u32 max_values,
                                long long* result_device) {
    unsigned char* const sc = (unsigned char*)sc_;
    const u32 n_tiles = lvs_tiles(len);
    hipLaunchKernelGGL(k_lv_setup, dim3((n_tiles + 255) / 256), dim3(256), 0, st, sc, n_tiles, (u32)width, len, out, max_values);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    LvsHead* const H = (LvsHead*)sc;
    LvTile* const order = (LvTile*)(sc + sizeof(LvsHead));
    u64* const pos_g = (u64*)(order + n_tiles);
    LvArgs a{&H->region, order, n_tiles, pos_g, pos_g + n_tiles, &H->ticket, nullptr, H->status, nullptr};
    e = launch_levels(st, runs, a, LV_T);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_lv_finish, dim3(1), dim3(1), 0, st, sc, n_tiles, result_device);
    return hipGetLastError();
}

}  // namespace pqd
#endif
