// Hybrid RLE / bit-packed level ENCODER on the device: the mirror of the decode path, byte for byte what
// rle.(*RLE) produces (internal/rle/rle.go:48-156: Write :48-67, writeOrAppendBitPackedRun :69-85,
// endPreviousBitPackedRun :87-96, writeRLERun :98-109, Bytes :140-156).
//
// What the reference's per-value state machine does, seen from a distance: at a group boundary, eight
// equal values open an RLE run that absorbs every following equal value; otherwise the next eight values
// (zero padded at the end of the input) become one bit-packed group, and consecutive groups share a
// header of at most 63 groups.  So the boundaries form a chain over the level positions,
//     next(i) = end of the equal run of i   if v[i..i+8) are all equal,      i + 8 otherwise,
// and, as in the decoder, a chain is turned into parallel work with tiles and exit tables.  A tile is
// EN_T positions.  The chain can enter a tile at its first 8 positions (a group stepped over the tile
// boundary) or where an RLE run that began earlier ends, which is one data-defined position per tile:
// 9 entries.
//   k_enc_lead     per tile: how far the equal run of the previous tile's last value reaches into it
//   k_enc_ext      per block: the same, continued across whole tiles (one backward pass over the tiles)
//   k_enc_exits    per tile: where the chain leaves the tile from each entry, elements passed (the tile is
//                  walked as 8 segments with 9 entries each at once, then the entries hop through them)
//   k_enc_hop      per block: true entry and element index of every tile
//   k_enc_mark     per tile: walks the true chain, one lane per segment, one record per element
//                  (position, RLE flag)
//   k_enc_chunks, k_enc_chunk_carry   byte offset of every chunk of 2048 elements (the bit-packed group index
//                  inside its run of at most 63 groups decides where headers go)
//   k_enc_write    per chunk: element offsets, then header / value bytes or the packed group per element;
//                  block length in front
#include "pqd_device.h"

namespace pqd {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_e __attribute__((aligned(1)));

// End (exclusive) of the equal run that position p of the tile belongs to, from the tile's change masks:
// bit k of s_mask[l] is set when v[tb + 32 l + k] != v[tb + 32 l + k + 1] (the last value of the block
// counts as changed).  s_next[l] = first lane >= l with a nonzero mask (64 if none).  Returns EN_T when
// the run reaches the end of the tile.
static __device__ __forceinline__ u32 run_end_in_tile(const u32* s_mask, const u32* s_next, u32 rel) {
    const u32 l = rel >> 5, k = rel & 31;
    const u32 m = s_mask[l] >> k;
    if (m) return rel + (u32)__builtin_ctz(m) + 1;
    if (l == 63) return EN_T;
    const u32 l2 = s_next[l + 1];
    if (l2 >= 64) return EN_T;
    return l2 * 32 + (u32)__builtin_ctz(s_mask[l2]) + 1;
}

// Builds the change masks of a tile in LDS (all 64 lanes of the wave take part).
static __device__ __forceinline__ void build_masks(const u8* v, u32 n, u32 tb, u32 lane, u32* s_mask, u32* s_next) {
    // my 32 values and the first value of the next lane
    const u32 p0 = tb + lane * 32;
    u32 w[9];
#pragma unroll
    for (int d = 0; d < 9; d++) {
        const u32 p = p0 + 4 * d;
        u32 x = 0;
        if (p + 4 <= n) x = *(const u32*)(v + p);  // v + tb is 4-byte aligned (tiles and blocks are)
        else for (u32 k = 0; p + k < n && k < 4; k++) x |= (u32)v[p + k] << (8 * k);
        w[d] = x;
    }
    u32 mask = 0;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const u32 a = (w[k >> 2] >> (8 * (k & 3))) & 0xFF, b = (w[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 0xFF;
        const u32 p = p0 + k;
        // the last value of the block ends its run; positions past the block never matter
        const bool ch = p + 1 >= n ? p + 1 == n : a != b;
        mask |= (u32)ch << k;
    }
    s_mask[lane] = mask;
    // first lane >= me with a nonzero mask: reverse min-scan
    u32 nx = mask ? lane : 64;
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_down(nx, d);
        if (lane + d < 64 && t < nx) nx = t;
    }
    s_next[lane] = nx;
}

}  // namespace

// lead[t] = number of leading positions of tile t (t > 0) equal to the last value of tile t - 1
__global__ __launch_bounds__(64) void k_enc_lead(const u8* __restrict__ src, const EncBlock* __restrict__ blocks,
                                                 const u32* __restrict__ tile_prefix, u32 n_blocks, u32* __restrict__ lead) {
    const u32 tile = blockIdx.x, lane = threadIdx.x;
    u32 lo = 0, hi = n_blocks;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (tile_prefix[mid] <= tile) lo = mid; else hi = mid; }
    const EncBlock B = blocks[lo];
    const u32 t = tile - tile_prefix[lo], tb = t * EN_T;
    if (t == 0) { if (lane == 0) lead[tile] = 0; return; }
    const u8* v = src + B.src_off;
    const u32 n = B.n, ref = v[tb - 1];
    u32 first = EN_T;  // first position (tile relative) that differs from ref
    for (u32 base = 0; base < EN_T && first == EN_T; base += 64 * 4) {
        const u32 p = tb + base + lane * 4;
        u32 x = ref * 0x01010101u;  // positions past the block compare equal; the result is clamped below
        if (p + 4 <= n) x = *(const u32*)(v + p);
        else for (u32 k = 0; k < 4 && p + k < n; k++) x = (x & ~(0xFFu << (8 * k))) | ((u32)v[p + k] << (8 * k));
        const u32 diff = x ^ (ref * 0x01010101u);
        u32 k = 4;
        if (diff) k = (u32)__builtin_ctz(diff) >> 3;
        const u64 b = __ballot(k < 4);
        if (b) {
            const int l = __ffsll((unsigned long long)b) - 1;
            first = base + (u32)l * 4 + __shfl(k, l);
        }
    }
    const u32 in_tile = n - tb < EN_T ? n - tb : EN_T;
    if (lane == 0) lead[tile] = first < in_tile ? first : in_tile;
}

// ext[t] = how far the equal run that reaches the end of tile t - 1 goes on from the start of tile t,
// across as many whole tiles as it covers: with nf = the first tile at or after t that the run does not
// cover entirely, ext[t] = (nf - t) * EN_T + lead[nf].  nf is a reverse min-scan over the block's tiles;
// one workgroup per block, batches of 1024 tiles from the end with a carry.
__global__ __launch_bounds__(1024) void k_enc_ext(const EncBlock* __restrict__ blocks, const u32* __restrict__ tile_prefix, u32 n_blocks,
                                                  const u32* __restrict__ lead, u32* __restrict__ ext) {
    __shared__ u32 s_w[16];
    const u32 b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 t0 = tile_prefix[b], nt = tile_prefix[b + 1] - t0;
    const u32 NONE = 0xFFFFFFFFu;
    u32 carry = NONE;  // first not-covered tile after the current batch
    for (u32 hi = nt; hi > 0; hi = hi > 1024 ? hi - 1024 : 0) {
        const u32 lo = hi > 1024 ? hi - 1024 : 0;
        const u32 t = lo + tid;
        const bool valid = t < hi;
        const u32 l = valid ? lead[t0 + t] : 0;
        u32 nf = valid && l < EN_T ? t : NONE;
        for (int d = 1; d < 64; d <<= 1) {
            const u32 o = __shfl_down(nf, d);
            if (lane + d < 64 && o < nf) nf = o;
        }
        if (lane == 0) s_w[wv] = nf;
        __syncthreads();
        u32 after = carry, batch_min = carry;
        for (u32 q = 0; q < 16; q++) {
            if (q > wv && s_w[q] < after) after = s_w[q];
            if (s_w[q] < batch_min) batch_min = s_w[q];
        }
        if (after < nf) nf = after;
        if (valid) ext[t0 + t] = nf == NONE ? (nt - t) * EN_T : (nf - t) * EN_T + lead[t0 + nf];
        carry = batch_min;
        __syncthreads();
    }
}

// One element step at block position p (tile relative rel): returns the next boundary (block position).
static __device__ __forceinline__ u32 enc_step(const u32* s_mask, const u32* s_next, const u32* __restrict__ ext_next, u32 tb,
                                               u32 rel, u32 n, bool& rle) {
    const u32 p = tb + rel;
    u32 re = run_end_in_tile(s_mask, s_next, rel);  // tile relative, EN_T = reaches the end of the tile
    u32 end = tb + re;
    if (re == EN_T && tb + EN_T < n) end = tb + EN_T + *ext_next;
    if (end > n) end = n;
    rle = p + 8 <= n && end - p >= 8;
    return rle ? end : p + 8;
}

struct EncExit { u32 pos; u32 cnt; };

// A tile is walked in EN_NSEG segments of EN_SEG positions at once.  What holds for a tile holds for a
// segment: the chain enters it at one of its first 8 positions or where the equal run that covers the
// position in front of it ends, so EN_E walkers per segment cover every way in, each takes at most
// EN_SEG / 8 steps, and the tile's exits follow by hopping through the segment table.  The table is kept
// for k_enc_mark: exit (tile relative, EN_T = beyond the tile, 0xFFFF = walker unused) | elements << 16.
constexpr u32 EN_SEG = 256, EN_NSEG = EN_T / EN_SEG, EN_WALKERS = EN_NSEG * EN_E;
constexpr u32 EN_NONE = 0xFFFFFFFFu;

__global__ __launch_bounds__(128) void k_enc_exits(const u8* __restrict__ src, const EncBlock* __restrict__ blocks,
                                                   const u32* __restrict__ tile_prefix, u32 n_blocks,
                                                   const u32* __restrict__ lead, const u32* __restrict__ ext,
                                                   EncExit* __restrict__ exits, u32* __restrict__ segs) {
    static_assert(EN_WALKERS <= 128 && EN_WALKERS == EN_SEG_RECS, "one thread per walker; the table the host sizes");
    __shared__ u32 s_mask[64], s_next[64];
    __shared__ u32 s_xpos[EN_WALKERS], s_xcnt[EN_WALKERS];
    const u32 tile = blockIdx.x, tid = threadIdx.x;
    u32 lo = 0, hi = n_blocks;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (tile_prefix[mid] <= tile) lo = mid; else hi = mid; }
    const EncBlock B = blocks[lo];
    const u32 t = tile - tile_prefix[lo], tb = t * EN_T, n = B.n;
    const u32 te = tb + EN_T, lim = te < n ? te : n;
    if (tid < 64) build_masks(src + B.src_off, n, tb, tid, s_mask, s_next);
    __syncthreads();
    const u32* ext_next = ext + tile + 1;  // the array has one spare entry
    const u32 ld = lead[tile];
    // tile entries 0..7: the first 8 positions; entry 8: where the run of the previous tile's last value
    // ends (below 8 it is one of the first; a run over the whole tile never enters it)
    const u32 lead_entry = ld >= 8 && ld < EN_T ? tb + ld : EN_NONE;
    if (tid < EN_WALKERS) {
        const u32 s = tid / EN_E, c = tid - s * EN_E;
        u32 p = EN_NONE;
        if (c < 8) p = tb + s * EN_SEG + c;
        else if (s == 0) p = lead_entry;
        else if (tb + s * EN_SEG - 1 < n) {
            const u32 re = run_end_in_tile(s_mask, s_next, s * EN_SEG - 1);
            u32 end = tb + re;
            if (re == EN_T && te < n) end = te + *ext_next;
            if (end > n) end = n;
            if (end >= tb + s * EN_SEG + 8) p = end;
        }
        const u32 start = p;
        const u32 seg_end = tb + (s + 1) * EN_SEG < lim ? tb + (s + 1) * EN_SEG : lim;
        u32 cnt = 0;
        while (p < seg_end) {
            bool rle;
            p = enc_step(s_mask, s_next, ext_next, tb, p - tb, n, rle);
            cnt++;
        }
        s_xpos[tid] = p; s_xcnt[tid] = cnt;
        const u32 rel = start == EN_NONE ? 0xFFFFu : p >= te ? EN_T : p - tb;
        segs[(u64)tile * EN_WALKERS + tid] = rel | (cnt << 16);
    }
    __syncthreads();
    if (tid < EN_E) {
        u32 p = tid < 8 ? tb + tid : lead_entry, cnt = 0;
        for (u32 h = 0; h < EN_NSEG && p < lim; h++) {  // every hop lands in a later segment
            const u32 rel = p - tb, s = rel / EN_SEG, off = rel - s * EN_SEG;
            const u32 w = s * EN_E + (off < 8 ? off : 8);  // off >= 8: the end of a run, walker 8
            p = s_xpos[w]; cnt += s_xcnt[w];
        }
        EncExit x;
        x.pos = p; x.cnt = cnt;
        exits[(u64)tile * EN_E + tid] = x;
    }
}

// True entry (block position, element index) of every tile; total elements of the block.  One workgroup
// per block.  When the block's exit table fits in LDS (a 1M-level page has 489 tiles) the walk has two
// levels, as in the decoder: groups of EN_HG tiles have the same 9 ways in as a tile (the first 8 positions,
// or the end of the run that covers the position in front of the group), so one thread per (group, entry)
// composes the group's exits, one thread follows the chain from group to group, and one thread per group
// writes the entries of its tiles.  Larger blocks are walked tile by tile by one thread.
constexpr u32 EN_HOP_LDS_TILES = 768, EN_HG = 8, EN_HOP_GROUPS = EN_HOP_LDS_TILES / EN_HG;
__global__ __launch_bounds__(256) void k_enc_hop(const EncBlock* __restrict__ blocks, const u32* __restrict__ tile_prefix, u32 n_blocks,
                                                 const u32* __restrict__ ext, const EncExit* __restrict__ exits,
                                                 EncExit* __restrict__ entries, u32* __restrict__ n_elems) {
    __shared__ EncExit s_ex[EN_HOP_LDS_TILES * EN_E];
    __shared__ EncExit s_gx[EN_HOP_GROUPS * EN_E], s_gent[EN_HOP_GROUPS];
    const u32 b = blockIdx.x, tid = threadIdx.x;
    const u32 n = blocks[b].n, t0 = tile_prefix[b], nt = tile_prefix[b + 1] - t0;
    if (nt > EN_HOP_LDS_TILES) {
        if (tid != 0) return;
        u32 p = 0, cnt = 0;
        while (p < n) {
            const u32 t = p / EN_T, off = p - t * EN_T;
            EncExit e;
            e.pos = p; e.cnt = cnt;
            entries[t0 + t] = e;
            const EncExit x = exits[(u64)t0 * EN_E + t * EN_E + (off < 8 ? off : 8)];  // off >= 8 is the run end: entry 8
            p = x.pos;
            cnt += x.cnt;
        }
        n_elems[b] = cnt;
        return;
    }
    const u32 ng = (nt + EN_HG - 1) / EN_HG;
    constexpr u32 GSPAN = EN_HG * EN_T;
    for (u32 i = tid; i < nt * EN_E; i += 256) s_ex[i] = exits[(u64)t0 * EN_E + i];
    for (u32 i = tid; i < ng; i += 256) s_gent[i].pos = EN_NONE;
    __syncthreads();
    for (u32 i = tid; i < ng * EN_E; i += 256) {
        const u32 g = i / EN_E, e = i - g * EN_E, gb = g * GSPAN;
        const u32 gend = gb + GSPAN < n ? gb + GSPAN : n;
        u32 p = EN_NONE, cnt = 0;
        if (e < 8) p = gb + e;
        else if (g > 0) {
            const u32 x = ext[t0 + g * EN_HG];  // how far the run in front of the group reaches into it (and beyond)
            if (x >= 8) p = gb + x;
        }
        while (p < gend) {
            const u32 t = p / EN_T, off = p - t * EN_T;
            const EncExit x = s_ex[t * EN_E + (off < 8 ? off : 8)];
            p = x.pos; cnt += x.cnt;
        }
        s_gx[i].pos = p; s_gx[i].cnt = cnt;
    }
    __syncthreads();
    if (tid == 0) {
        u32 p = 0, cnt = 0;
        while (p < n) {
            const u32 g = p / GSPAN, off = p - g * GSPAN;
            s_gent[g].pos = p; s_gent[g].cnt = cnt;
            const EncExit x = s_gx[g * EN_E + (off < 8 ? off : 8)];
            p = x.pos; cnt += x.cnt;
        }
        n_elems[b] = cnt;
    }
    __syncthreads();
    for (u32 g = tid; g < ng; g += 256) {
        u32 p = s_gent[g].pos, cnt = s_gent[g].cnt;
        const u32 gend = (g + 1) * GSPAN < n ? (g + 1) * GSPAN : n;
        while (p < gend) {  // EN_NONE: the chain steps over the whole group
            const u32 t = p / EN_T, off = p - t * EN_T;
            EncExit e;
            e.pos = p; e.cnt = cnt;
            entries[t0 + t] = e;
            const EncExit x = s_ex[t * EN_E + (off < 8 ? off : 8)];
            p = x.pos; cnt += x.cnt;
        }
    }
}

// Element records: position | RLE flag in bit 31 (levels per block stay below 2^31).  Lane 0 hops from the
// tile's true entry through the segment table of k_enc_exits, which yields where the true chain enters every
// segment and how many elements precede it; then one lane per segment walks and writes its records.
__global__ __launch_bounds__(64) void k_enc_mark(const u8* __restrict__ src, const EncBlock* __restrict__ blocks,
                                                 const u32* __restrict__ tile_prefix, u32 n_blocks,
                                                 const u32* __restrict__ ext, const EncExit* __restrict__ entries,
                                                 const u32* __restrict__ segs, u32* __restrict__ elems) {
    __shared__ u32 s_mask[64], s_next[64];
    __shared__ u32 s_seg[EN_WALKERS], s_entp[EN_NSEG], s_entc[EN_NSEG];
    const u32 tile = blockIdx.x, lane = threadIdx.x;
    const EncExit ent = entries[tile];
    if (ent.pos == EN_NONE) return;  // the true chain has no boundary in this tile
    u32 lo = 0, hi = n_blocks;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (tile_prefix[mid] <= tile) lo = mid; else hi = mid; }
    const EncBlock B = blocks[lo];
    const u32 t = tile - tile_prefix[lo], tb = t * EN_T, n = B.n, te = tb + EN_T, lim = te < n ? te : n;
    build_masks(src + B.src_off, n, tb, lane, s_mask, s_next);
    for (u32 i = lane; i < EN_WALKERS; i += 64) s_seg[i] = segs[(u64)tile * EN_WALKERS + i];
    if (lane < EN_NSEG) s_entp[lane] = EN_NONE;
    __syncthreads();
    if (lane == 0) {
        u32 p = ent.pos, cnt = 0;
        for (u32 h = 0; h < EN_NSEG && p < lim; h++) {
            const u32 rel = p - tb, s = rel / EN_SEG, off = rel - s * EN_SEG;
            s_entp[s] = p; s_entc[s] = cnt;
            const u32 r = s_seg[s * EN_E + (off < 8 ? off : 8)], x = r & 0xFFFFu;
            cnt += r >> 16;
            p = x >= EN_T ? EN_NONE : tb + x;
        }
    }
    __syncthreads();
    if (lane >= EN_NSEG) return;
    u32 p = s_entp[lane];
    if (p == EN_NONE) return;
    const u32 seg_end = tb + (lane + 1) * EN_SEG < lim ? tb + (lane + 1) * EN_SEG : lim;
    u32* out = elems + B.elem_base + ent.cnt + s_entc[lane];
    u32 i = 0;
    while (p < seg_end) {
        bool rle;
        const u32 q = enc_step(s_mask, s_next, ext + tile + 1, tb, p - tb, n, rle);
        out[i++] = p | (rle ? 0x80000000u : 0u);
        p = q;
    }
}

// Layout: the byte offset of every element, in chunks of EN_LC elements so that the whole device takes part.
//   bit-packed element e: k = e - 1 - (index of the last RLE element before e); it opens a header when
//   k % 63 == 0 (writeOrAppendBitPackedRun: a new run after an RLE run or after 63 groups)
//   sizes: RLE = leb128(count << 1) + one value byte unless width is 0; bit-packed = width bytes (+1 header)
// A chunk on its own knows everything except the headers of its leading bit-packed elements (their k counts
// from an RLE element of an earlier chunk), and those follow in closed form from how many they are and the k of
// the first.  So: k_enc_chunks sums each chunk without them, k_enc_chunk_carry scans the chunks of a block
// (last RLE element so far, bytes so far), and k_enc_write redoes the chunk's scan with both known and writes.
struct EncChunk {
    u32 sum;   // bytes of the chunk without the headers of its leading bit-packed elements
    u32 head;  // number of those elements
    int last;  // k_enc_chunks: last RLE element of the chunk (-1 none); after the carry: the last one BEFORE the chunk
    u32 base;  // after the carry: byte offset of the chunk
};
static __device__ __forceinline__ u32 mult63(u32 x) { return (x + 62) / 63; }  // multiples of 63 in [0, x)

// One chunk, 256 threads: wave wv, row j, lane l holds element c0 + wv * 512 + j * 64 + l, so loads are
// coalesced and a wave's rows are consecutive.  Yields per element its record, the record after it, its k
// and its size.  FINAL = false: `carry` is unknown, leading bit-packed elements get size `width` and are
// counted in head.  Returns the chunk's last RLE element (-1: none).
template <bool FINAL>
static __device__ __forceinline__ int chunk_sizes(const u32* __restrict__ el, u32 ne, u32 c0, u32 bn, u32 width, int carry,
                                                  u32 tid, int* s_last, u32 (&rec)[8], u32 (&nxt)[8], u32 (&kk)[8],
                                                  u32 (&size)[8], u32& head) {
    const u32 lane = tid & 63, wv = tid >> 6;
    const u64 lt = (1ull << lane) - 1;
    const u32 bw = width ? 1u : 0u;
    int before[8];
    int rowcarry = -1;
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
        const u32 r0 = c0 + wv * 512 + j * 64, e = r0 + lane;
        rec[j] = e < ne ? el[e] : 0;
        nxt[j] = e + 1 < ne ? el[e + 1] : bn;
        // elements of a row are consecutive: the last RLE element before me is the highest set bit below my lane
        const u64 b = __ballot((rec[j] & 0x80000000u) != 0);
        const u64 m = b & lt;
        before[j] = m ? (int)(r0 + 63 - (u32)__builtin_clzll(m)) : rowcarry;
        if (b) rowcarry = (int)(r0 + 63 - (u32)__builtin_clzll(b));
    }
    if (lane == 0) s_last[wv] = rowcarry;
    __syncthreads();
    int cross = -1, chunk_last = -1;
    for (u32 q = 0; q < 4; q++) {
        if (q < wv && s_last[q] > cross) cross = s_last[q];
        if (s_last[q] > chunk_last) chunk_last = s_last[q];
    }
    head = 0;
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
        const u32 e = c0 + wv * 512 + j * 64 + lane;
        int bf = before[j] > cross ? before[j] : cross;
        size[j] = 0; kk[j] = 0;
        if (e < ne) {
            if (rec[j] & 0x80000000u) {
                const u32 h = ((nxt[j] & 0x7FFFFFFFu) - (rec[j] & 0x7FFFFFFFu)) << 1;  // a run is shorter than 2^31 values
                size[j] = (h < (1u << 7) ? 1u : h < (1u << 14) ? 2u : h < (1u << 21) ? 3u : h < (1u << 28) ? 4u : 5u) + bw;
            } else if (!FINAL && bf < 0) {
                size[j] = width;
                head++;
            } else {
                if (bf < 0) bf = carry;
                kk[j] = e - 1 - (u32)bf;  // bf == -1: e
                size[j] = width + (kk[j] % 63 == 0 ? 1u : 0u);
            }
        }
    }
    return chunk_last;
}

constexpr u32 EN_LC = EN_LAYOUT_CHUNK;
static __device__ __forceinline__ u32 find_block(const u32* __restrict__ prefix, u32 n_blocks, u32 x) {
    u32 lo = 0, hi = n_blocks;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (prefix[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(256) void k_enc_chunks(const EncBlock* __restrict__ blocks, const u32* __restrict__ lchunk_prefix,
                                                    u32 n_blocks, const u32* __restrict__ n_elems, const u32* __restrict__ elems,
                                                    EncChunk* __restrict__ chunks) {
    __shared__ int s_last[4];
    __shared__ u32 s_sum[4], s_head[4];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 blk = find_block(lchunk_prefix, n_blocks, blockIdx.x);
    const u32 c0 = (blockIdx.x - lchunk_prefix[blk]) * EN_LC, ne = n_elems[blk];
    if (c0 >= ne) return;
    const EncBlock B = blocks[blk];
    u32 rec[8], nxt[8], kk[8], size[8], head;
    const int last = chunk_sizes<false>(elems + B.elem_base, ne, c0, B.n, B.width, -1, tid, s_last, rec, nxt, kk, size, head);
    u32 sum = 0;
#pragma unroll
    for (u32 j = 0; j < 8; j++) sum += size[j];
    for (int d = 32; d > 0; d >>= 1) { sum += __shfl_xor(sum, d); head += __shfl_xor(head, d); }
    if (lane == 0) { s_sum[wv] = sum; s_head[wv] = head; }
    __syncthreads();
    if (tid == 0) {
        EncChunk c;
        c.sum = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        c.head = s_head[0] + s_head[1] + s_head[2] + s_head[3];
        c.last = last; c.base = 0;
        chunks[blockIdx.x] = c;
    }
}

// One wave per block: scans the block's chunks, 64 at a time.
__global__ __launch_bounds__(64) void k_enc_chunk_carry(const u32* __restrict__ lchunk_prefix, const u32* __restrict__ n_elems,
                                                        EncChunk* __restrict__ chunks, u32* __restrict__ out_len) {
    const u32 b = blockIdx.x, lane = threadIdx.x;
    const u32 ne = n_elems[b], nc = (ne + EN_LC - 1) / EN_LC;
    EncChunk* ch = chunks + lchunk_prefix[b];
    int carry_before = -1;
    u32 carry_off = 0;
    for (u32 cb = 0; cb < nc; cb += 64) {
        const u32 c = cb + lane;
        const bool valid = c < nc;
        EncChunk x;
        x.sum = 0; x.head = 0; x.last = -1; x.base = 0;
        if (valid) x = ch[c];
        int incl = x.last;
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= (u32)d && t > incl) incl = t;
        }
        int before = __shfl_up(incl, 1);
        if (lane == 0) before = -1;
        if (carry_before > before) before = carry_before;
        // the leading bit-packed elements of the chunk: the first has k0, every multiple of 63 is a header
        const u32 k0 = c * EN_LC - 1 - (u32)before;
        const u32 total = valid ? x.sum + mult63(k0 + x.head) - mult63(k0) : 0;
        u32 incl_s = total;
        for (int d = 1; d < 64; d <<= 1) {
            const u32 t = __shfl_up(incl_s, d);
            if (lane >= (u32)d) incl_s += t;
        }
        if (valid) { ch[c].last = before; ch[c].base = carry_off + incl_s - total; }
        const int top = __shfl(incl, 63);
        if (top > carry_before) carry_before = top;
        carry_off += __shfl(incl_s, 63);
    }
    if (lane == 0) out_len[b] = carry_off;
}

// bitpack.Pack for one group (internal/bitpack/bitpack.go:20-74, generated by cmd/bitpackgen: value j
// occupies bits [j*w, (j+1)*w) of the group's w bytes, LSB first)
static __device__ __forceinline__ void pack_group(const u8* v, u32 p, u32 n, u32 w, u8* out) {
    u64 acc = 0;
    const u64 mask = (1ull << w) - 1;
    for (u32 j = 0; j < 8; j++) {
        const u64 x = p + j < n ? v[p + j] : 0;  // the last group is zero padded
        acc |= (x & mask) << (j * w);
    }
    for (u32 b = 0; b < w; b++) out[b] = (u8)(acc >> (8 * b));
}

__global__ __launch_bounds__(256) void k_enc_write(const u8* __restrict__ src, const EncBlock* __restrict__ blocks,
                                                   const u32* __restrict__ lchunk_prefix, u32 n_blocks,
                                                   const u32* __restrict__ n_elems, const u32* __restrict__ elems,
                                                   const EncChunk* __restrict__ chunks, const u32* __restrict__ out_len) {
    __shared__ int s_last[4];
    __shared__ u32 s_sum[4];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 blk = find_block(lchunk_prefix, n_blocks, blockIdx.x);
    const u32 c0 = (blockIdx.x - lchunk_prefix[blk]) * EN_LC, ne = n_elems[blk];
    const EncBlock B = blocks[blk];
    u8* out = B.out;
    if (c0 == 0 && tid == 0) {  // int32 LE length in front of the runs (rle.go:153-155); also for an empty block
        const u32 L = out_len[blk];
        out[0] = (u8)L; out[1] = (u8)(L >> 8); out[2] = (u8)(L >> 16); out[3] = (u8)(L >> 24);
    }
    if (c0 >= ne) return;
    const EncChunk ch = chunks[blockIdx.x];
    u32 rec[8], nxt[8], kk[8], size[8], head;
    chunk_sizes<true>(elems + B.elem_base, ne, c0, B.n, B.width, ch.last, tid, s_last, rec, nxt, kk, size, head);
    // offsets: sum-scan in element order (rows of a wave, then the waves)
    u32 off[8], rowbase = 0;
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
        u32 incl = size[j];
        for (int d = 1; d < 64; d <<= 1) {
            const u32 t = __shfl_up(incl, d);
            if (lane >= (u32)d) incl += t;
        }
        off[j] = rowbase + incl - size[j];
        rowbase += __shfl(incl, 63);
    }
    if (lane == 0) s_sum[wv] = rowbase;
    __syncthreads();
    u32 wbase = ch.base;
    for (u32 q = 0; q < wv; q++) wbase += s_sum[q];
    const u8* v = src + B.src_off;
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
        const u32 e = c0 + wv * 512 + j * 64 + lane;
        if (e >= ne) continue;
        const u32 p = rec[j] & 0x7FFFFFFFu, o0 = wbase + off[j];
        u8* o = out + 4 + o0;
        if (rec[j] & 0x80000000u) {
            u32 h = ((nxt[j] & 0x7FFFFFFFu) - p) << 1;
            while (h >= 0x80) { *o++ = (u8)(h | 0x80); h >>= 7; }
            *o++ = (u8)h;
            if (B.width) *o = v[p];
        } else {
            const u32 m = kk[j] % 63;
            if (m == 0) o++;  // the header byte of the run this group opens
            pack_group(v, p, B.n, B.width, o);
            // the last group of a bit-packed run writes the run's header, groups << 1 | 1, which sits in front of
            // the run's first group: m groups of `width` bytes and the header byte itself back from here
            const bool last = m == 62 || e + 1 == ne || (nxt[j] & 0x80000000u);
            if (last) out[4 + o0 - (m ? m * B.width + 1 : 0)] = (u8)(((m + 1) << 1) | 1);
        }
    }
}

// =====================================================================================
// Page payloads (OptionalField.DoWrite / RequiredField.DoWrite, fields.go:95-111, 232-271): repetition
// levels, definition levels, then the PLAIN values, back to back.
//   k_encp_place   sizes of the three parts of every page -> payload offsets (scan over the pages), and the
//                  destinations of the level blocks
//   k_encp_values  PLAIN values: fixed width = the dense little-endian values as they are
//                  (template_required.go:37-49, template_optional.go:96-114: binary.Write), bools 8 per
//                  byte LSB first (template_bool.go), strings int32 length + bytes (template_string.go:62-78)
//   k_encp_stats   min / max of the page's values the way the generated stats types keep them: min starts at
//                  the type's maximum, max starts at ZERO (template_required.go:77-96, template_optional.go:
//                  79-113), comparisons in the Go type (signed, unsigned, float)
// =====================================================================================
__global__ __launch_bounds__(1024) void k_encp_place(EncPage* __restrict__ pages, u32 n_pages, EncBlock* __restrict__ blocks,
                                                     const u32* __restrict__ out_len, const u8* __restrict__ src,
                                                     u8* __restrict__ out_base) {
    __shared__ u64 s_sum[16];
    const u32 tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 per = (n_pages + 1023) / 1024;
    const u32 a = tid * per < n_pages ? tid * per : n_pages, b = a + per < n_pages ? a + per : n_pages;
    u64 mine = 0;
    for (u32 i = a; i < b; i++) {
        EncPage P = pages[i];
        P.rep_bytes = P.rep_block != 0xFFFFFFFFu ? out_len[P.rep_block] + 4 : 0;
        P.def_bytes = P.def_block != 0xFFFFFFFFu ? out_len[P.def_block] + 4 : 0;
        u64 vb;
        if (P.width) vb = P.n_values * P.width;
        else if (P.phys_type == PT_BOOLEAN) vb = (P.n_values + 7) / 8;
        else {
            const int* offs = (const int*)(src + P.offsets_off);
            vb = 4 * P.n_values + (u64)(u32)(offs[P.n_values] - offs[0]);
        }
        P.val_bytes = vb;
        pages[i].rep_bytes = P.rep_bytes; pages[i].def_bytes = P.def_bytes; pages[i].val_bytes = vb;
        mine += (u64)P.rep_bytes + P.def_bytes + vb;
    }
    u64 incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const u64 t = __shfl_up(incl, d);
        if (lane >= (u32)d) incl += t;
    }
    if (lane == 63) s_sum[wv] = incl;
    __syncthreads();
    u64 run = incl - mine;
    for (u32 q = 0; q < wv; q++) run += s_sum[q];
    for (u32 i = a; i < b; i++) {
        EncPage P = pages[i];
        pages[i].payload_off = run;
        if (P.rep_block != 0xFFFFFFFFu) blocks[P.rep_block].out = out_base + run;
        if (P.def_block != 0xFFFFFFFFu) blocks[P.def_block].out = out_base + run + P.rep_bytes;
        run += (u64)P.rep_bytes + P.def_bytes + P.val_bytes;
    }
}

constexpr u32 ENP_CHUNK = ENC_VALUE_CHUNK;
__global__ __launch_bounds__(256) void k_encp_values(const EncPage* __restrict__ pages, const u32* __restrict__ chunk_prefix,
                                                     u32 n_pages, const u8* __restrict__ src, u8* __restrict__ out_base) {
    const u32 blk = blockIdx.x, tid = threadIdx.x;
    u32 lo = 0, hi = n_pages;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (chunk_prefix[mid] <= blk) lo = mid; else hi = mid; }
    const EncPage P = pages[lo];
    const u64 c0 = (u64)(blk - chunk_prefix[lo]) * ENP_CHUNK;
    u8* dst = out_base + P.payload_off + P.rep_bytes + P.def_bytes;
    if (P.width) {
        const u64 n = P.n_values * P.width;
        const u8* s = src + P.values_off;
        const u64 c1 = c0 + ENP_CHUNK < n ? c0 + ENP_CHUNK : n;
        for (u64 i = c0 + tid * 16; i < c1; i += 256 * 16) {
            if (i + 16 <= c1) *(u32x4_e*)(dst + i) = *(const u32x4_e*)(s + i);
            else for (u64 k = i; k < c1; k++) dst[k] = s[k];
        }
    } else if (P.phys_type == PT_BOOLEAN) {
        // the chunk counts output bytes; 8 one-byte bools each
        const u64 nb = (P.n_values + 7) / 8;
        const u8* s = src + P.values_off;
        const u64 c1 = c0 + ENP_CHUNK < nb ? c0 + ENP_CHUNK : nb;
        for (u64 i = c0 + tid; i < c1; i += 256) {
            u32 x = 0;
            for (u32 j = 0; j < 8 && i * 8 + j < P.n_values; j++) x |= (u32)(s[i * 8 + j] != 0) << j;
            dst[i] = (u8)x;
        }
    } else {
        // the chunk counts strings here: ENP_CHUNK / 16 of them per workgroup, one thread per string
        const int* offs = (const int*)(src + P.offsets_off);
        const u8* chars = src + P.values_off;
        const u64 per = ENP_CHUNK / 16, s0 = (u64)(blk - chunk_prefix[lo]) * per;
        const u64 s1 = s0 + per < P.n_values ? s0 + per : P.n_values;
        const int base = offs[0];
        for (u64 i = s0 + tid; i < s1; i += 256) {
            const u32 a = (u32)(offs[i] - base), len = (u32)(offs[i + 1] - offs[i]);
            u8* d = dst + 4 * i + a;
            d[0] = (u8)len; d[1] = (u8)(len >> 8); d[2] = (u8)(len >> 16); d[3] = (u8)(len >> 24);
            const u8* c = chars + offs[i];
            for (u32 k = 0; k < len; k++) d[4 + k] = c[k];
        }
    }
}

// statistics in two steps: every 64 KiB chunk of a page's values reduces to a partial (min, max), then one
// wave per page folds its partials.  Both start from the stats types' initial values (type maximum, zero).
template <typename T>
static __device__ __forceinline__ void stats_init(u32 kind, T& mn, T& mx) {
    mx = (T)0;
    if (sizeof(T) == 4) { const u32 b = kind == 2 ? 0x7F7FFFFFu : kind == 1 ? 0xFFFFFFFFu : 0x7FFFFFFFu; memcpy(&mn, &b, 4); }
    else { const u64 b = kind == 2 ? 0x7FEFFFFFFFFFFFFFull : kind == 1 ? ~0ull : 0x7FFFFFFFFFFFFFFFull; memcpy(&mn, &b, 8); }
}
template <typename T>
static __device__ __forceinline__ void stats_wave(T& mn, T& mx) {
    for (int d = 32; d > 0; d >>= 1) {
        T a, b;
        if (sizeof(T) == 4) {
            u32 t, u; memcpy(&t, &mn, 4); memcpy(&u, &mx, 4);
            t = __shfl_down(t, d); u = __shfl_down(u, d);
            memcpy(&a, &t, 4); memcpy(&b, &u, 4);
        } else {
            u64 t, u; memcpy(&t, &mn, 8); memcpy(&u, &mx, 8);
            t = __shfl_down(t, d); u = __shfl_down(u, d);
            memcpy(&a, &t, 8); memcpy(&b, &u, 8);
        }
        if (a < mn) mn = a;
        if (b > mx) mx = b;
    }
}
template <typename T>
static __device__ void stats_chunk(const EncPage& P, const u8* src, u64 c0, u64* part) {
    __shared__ unsigned long long s_min[4], s_max[4];
    const T* v = (const T*)(src + P.values_off);
    const u64 i0 = c0 / sizeof(T), i1 = i0 + ENC_VALUE_CHUNK / sizeof(T) < P.n_values ? i0 + ENC_VALUE_CHUNK / sizeof(T) : P.n_values;
    T mn, mx;
    stats_init<T>(P.stats_kind, mn, mx);
    for (u64 i = i0 + threadIdx.x; i < i1; i += 256) {
        const T x = v[i];
        if (x < mn) mn = x;
        if (x > mx) mx = x;
    }
    stats_wave<T>(mn, mx);
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { unsigned long long t = 0, u = 0; memcpy(&t, &mn, sizeof(T)); memcpy(&u, &mx, sizeof(T)); s_min[wv] = t; s_max[wv] = u; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (u32 q = 1; q < 4; q++) {
            T a, b;
            memcpy(&a, &s_min[q], sizeof(T)); memcpy(&b, &s_max[q], sizeof(T));
            if (a < mn) mn = a;
            if (b > mx) mx = b;
        }
        u64 t = 0, u = 0;
        memcpy(&t, &mn, sizeof(T)); memcpy(&u, &mx, sizeof(T));
        part[0] = t; part[1] = u;
    }
}
template <typename T>
static __device__ void stats_fold(const EncPage& P, const u64* part, u32 n_part, EncPage* out) {
    T mn, mx;
    stats_init<T>(P.stats_kind, mn, mx);
    for (u32 i = threadIdx.x; i < n_part; i += 64) {
        T a, b;
        memcpy(&a, &part[2 * i], sizeof(T)); memcpy(&b, &part[2 * i + 1], sizeof(T));
        if (a < mn) mn = a;
        if (b > mx) mx = b;
    }
    stats_wave<T>(mn, mx);
    if (threadIdx.x == 0) {
        u64 t = 0, u = 0;
        memcpy(&t, &mn, sizeof(T)); memcpy(&u, &mx, sizeof(T));
        out->min_bits = t; out->max_bits = u;
    }
}

#define STATS_DISPATCH(FN, ...)                                              \
    do {                                                                      \
        if (P.width == 4) {                                                   \
            if (P.stats_kind == 0) FN<int>(__VA_ARGS__);                      \
            else if (P.stats_kind == 1) FN<u32>(__VA_ARGS__);                 \
            else FN<float>(__VA_ARGS__);                                      \
        } else {                                                              \
            if (P.stats_kind == 0) FN<long long>(__VA_ARGS__);                \
            else if (P.stats_kind == 1) FN<u64>(__VA_ARGS__);                 \
            else FN<double>(__VA_ARGS__);                                     \
        }                                                                     \
    } while (0)

// stats_kind: 0 signed, 1 unsigned, 2 floating point
__global__ __launch_bounds__(256) void k_encp_stats_chunks(const EncPage* __restrict__ pages, const u32* __restrict__ chunk_prefix,
                                                           u32 n_pages, const u8* __restrict__ src, u64* __restrict__ partial) {
    const u32 blk = blockIdx.x;
    u32 lo = 0, hi = n_pages;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (chunk_prefix[mid] <= blk) lo = mid; else hi = mid; }
    const EncPage P = pages[lo];
    if (!P.width) return;
    const u64 c0 = (u64)(blk - chunk_prefix[lo]) * ENC_VALUE_CHUNK;
    STATS_DISPATCH(stats_chunk, P, src, c0, partial + 2 * (u64)blk);
}
__global__ __launch_bounds__(64) void k_encp_stats_pages(EncPage* __restrict__ pages, const u32* __restrict__ chunk_prefix,
                                                         const u64* __restrict__ partial) {
    const EncPage P = pages[blockIdx.x];
    if (!P.width) return;
    const u32 c0 = chunk_prefix[blockIdx.x], c1 = chunk_prefix[blockIdx.x + 1];
    STATS_DISPATCH(stats_fold, P, partial + 2 * (u64)c0, c1 - c0, pages + blockIdx.x);
}

// ---- launchers -------------------------------------------------------------------------------
#define ENC_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

// layout: everything up to the size of every block (out_len) and the byte offset of every element
hipError_t launch_rle_encode_layout(hipStream_t st, const u8* src, const EncArgs& a) {
    if (a.n_blocks == 0) return hipSuccess;
    if (a.n_tiles) {
        hipLaunchKernelGGL(k_enc_lead, dim3(a.n_tiles), dim3(64), 0, st, src, a.blocks, a.tile_prefix, a.n_blocks, a.lead);
        ENC_CHECK();
        hipLaunchKernelGGL(k_enc_ext, dim3(a.n_blocks), dim3(1024), 0, st, a.blocks, a.tile_prefix, a.n_blocks, a.lead, a.ext);
        ENC_CHECK();
        hipLaunchKernelGGL(k_enc_exits, dim3(a.n_tiles), dim3(128), 0, st, src, a.blocks, a.tile_prefix, a.n_blocks, a.lead, a.ext,
                           (EncExit*)a.exits, a.segs);
        ENC_CHECK();
    }
    hipLaunchKernelGGL(k_enc_hop, dim3(a.n_blocks), dim3(256), 0, st, a.blocks, a.tile_prefix, a.n_blocks, a.ext,
                       (const EncExit*)a.exits, (EncExit*)a.entries, a.n_elems);
    ENC_CHECK();
    if (a.n_tiles) {
        hipLaunchKernelGGL(k_enc_mark, dim3(a.n_tiles), dim3(64), 0, st, src, a.blocks, a.tile_prefix, a.n_blocks, a.ext,
                           (const EncExit*)a.entries, a.segs, a.elems);
        ENC_CHECK();
    }
    hipLaunchKernelGGL(k_enc_chunks, dim3(a.n_lchunks), dim3(256), 0, st, a.blocks, a.lchunk_prefix, a.n_blocks, a.n_elems, a.elems,
                       (EncChunk*)a.chunks);
    ENC_CHECK();
    hipLaunchKernelGGL(k_enc_chunk_carry, dim3(a.n_blocks), dim3(64), 0, st, a.lchunk_prefix, a.n_elems, (EncChunk*)a.chunks, a.out_len);
    ENC_CHECK();
    return hipSuccess;
}
// write: the bytes, to EncBlock.out of every block
hipError_t launch_rle_encode_write(hipStream_t st, const u8* src, const EncArgs& a) {
    if (a.n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_enc_write, dim3(a.n_lchunks), dim3(256), 0, st, src, a.blocks, a.lchunk_prefix, a.n_blocks, a.n_elems,
                       a.elems, (const EncChunk*)a.chunks, a.out_len);
    ENC_CHECK();
    return hipSuccess;
}
hipError_t launch_rle_encode(hipStream_t st, const u8* src, const EncArgs& a) {
    hipError_t e = launch_rle_encode_layout(st, src, a);
    return e != hipSuccess ? e : launch_rle_encode_write(st, src, a);
}

hipError_t launch_encode_pages(hipStream_t st, const u8* src, EncPage* pages, const u32* chunk_prefix, u32 n_pages, u32 n_chunks,
                               EncBlock* blocks, const u32* out_len, u8* out_base, u64* stats_partial) {
    if (n_pages == 0) return hipSuccess;
    hipLaunchKernelGGL(k_encp_place, dim3(1), dim3(1024), 0, st, pages, n_pages, blocks, out_len, src, out_base);
    ENC_CHECK();
    if (n_chunks) {
        hipLaunchKernelGGL(k_encp_values, dim3(n_chunks), dim3(256), 0, st, pages, chunk_prefix, n_pages, src, out_base);
        ENC_CHECK();
    }
    if (n_chunks) {
        hipLaunchKernelGGL(k_encp_stats_chunks, dim3(n_chunks), dim3(256), 0, st, pages, chunk_prefix, n_pages, src, stats_partial);
        ENC_CHECK();
    }
    hipLaunchKernelGGL(k_encp_stats_pages, dim3(n_pages), dim3(64), 0, st, pages, chunk_prefix, stats_partial);
    ENC_CHECK();
    return hipSuccess;
}

}  // namespace pqd
