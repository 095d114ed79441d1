// bin_kernels.hip -- tile binning BY COUNTING: the per-tile splat lists, in depth order, without a sort over the R instances.
//
// The public algorithm duplicates every Gaussian into one (tile, depth) key per touched tile and radix-sorts the R keys; rounds
// 1-3 of this library sorted the Gaussians by depth first (N keys) and then stable-sorted the R (tile, id) pairs by tile id in two
// radix passes -- 9 launches and ~200 MB of key/value traffic at configs[2].  But "stable sort by tile id of a sequence that is
// already in depth order" is a COUNTING sort whose histogram has one bin per tile, and on CDNA4 a workgroup's LDS (160 KB) holds
// that whole histogram (T = 5 440 tiles at 1352 x 1014: 22 KB):
//
//   A  gp_bin_count_kernel     block b owns the b-th chunk of the depth-ordered Gaussians: counts its instances per tile in LDS --
//                              per QUARTER of the chunk, in packed 16-bit counters -- and writes the row hist[b][0..T), the four
//                              quarter rows for kernel C (and adds its instance count into the R slots)
//   B  gp_bin_scan_kernel      per tile: exclusive prefix of the counts over the blocks (in place) + the tile's total
//   C  gp_bin_scatter_kernel   every block: tile_start = exclusive scan of the totals (block 0 also writes `ranges` and the status
//                              word), the quarter rows prefixed over its four waves (wave w owns quarter w), then walks its chunk
//                              and stores every instance's Gaussian id at
//                              tile_start[tile] + (instances of that tile in earlier blocks / waves / steps / lanes)
//
// Order inside a tile = block order, then wave order (wave w owns the w-th quarter of the chunk), then the 64-instance steps of the
// wave, then lane order, with the ranks of equal tiles inside a step from wave64 match-any ballots: exactly the depth order, so the
// result is bit-identical to the stable sort it replaces (tests/test_gpu_raster.py compares the whole point_list with the
// oracle's).  No key array exists any more: instances are never materialised, only their final slots are written.
// Traffic: 2 x 12 B per Gaussian + 4 B per instance + 3 x 4 B x T x NB of histogram (11 MB at NB = 489) instead of 48 B per instance.
#include "gp_common.h"
#include "raster_kernels.h"
#include <stdlib.h>

#define BIN_THREADS 256
#define BIN_WAVES (BIN_THREADS / GP_WAVE)
#define BIN_SEGS 16                // gp_bin_scan_kernel: 16 waves, each sums a 16th of the blocks
#define BIN_SEG_MAX 32             // ... with up to 32 counters per thread in registers (NB <= 512)

// inclusive wave64 scans on the DPP network (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, row_bcast 15 / 31 across them): no
// LDS round trips -- __shfl_up is a ds_bpermute, six DEPENDENT trips per scan, and these kernels run at one to three waves per
// SIMD, where nothing hides them
#define BIN_DPP(x, ctrl, rmask) __builtin_amdgcn_update_dpp(0, (x), (ctrl), (rmask), 0xf, false)
__device__ __forceinline__ int bin_scan_add(int x) {
    x += BIN_DPP(x, 0x111, 0xf); x += BIN_DPP(x, 0x112, 0xf); x += BIN_DPP(x, 0x114, 0xf); x += BIN_DPP(x, 0x118, 0xf);
    x += BIN_DPP(x, 0x142, 0xa); x += BIN_DPP(x, 0x143, 0xc);
    return x;
}
__device__ __forceinline__ int bin_scan_max(int x) {          // (values >= 0)
    x = max(x, BIN_DPP(x, 0x111, 0xf)); x = max(x, BIN_DPP(x, 0x112, 0xf)); x = max(x, BIN_DPP(x, 0x114, 0xf)); x = max(x, BIN_DPP(x, 0x118, 0xf));
    x = max(x, BIN_DPP(x, 0x142, 0xa)); x = max(x, BIN_DPP(x, 0x143, 0xc));
    return x;
}

// Instances of the 64 Gaussians of a wave, 128 at a time.  Lane g holds Gaussian g's run (cnt = w * h tiles of the rectangle at
// (minx, miny), row-major); the runs are laid end to end (exclusive scan of cnt) and an iteration covers instances [b, b + 128) as
// two steps of 64.  Which Gaussian owns instance j: every run that begins inside the iteration -- or reaches into it from before --
// marks its first slot in LDS, and an inclusive max-scan over the lanes spreads the marks (a binary search over the run starts, as
// gp_duplicate_kernel does it, is six dependent LDS trips; here there are two per iteration, and the two steps' chains overlap).
// The owner's rectangle comes back from the per-wave table s_own with one 16-byte read.
// f(valid, tile, id) is called once per step by all lanes, steps in order.  BinWave: this wave's LDS scratch (marks zero on entry
// and on exit).
struct BinWave {
    int4 own[64];            // (run start, first tile x, first tile y, tiles per row)
    uint32_t id[64];
    int mark[128];
};
template <bool WITH_ID, class F>
__device__ __forceinline__ void bin_walk(int lane, BinWave& sw, int cnt, int minx, int miny, int w, uint32_t id, int gx, F&& f) {
    const int incl = bin_scan_add(cnt);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    const int start = incl - cnt;
    sw.own[lane] = make_int4(start, minx, miny, w);
    if (WITH_ID) sw.id[lane] = id;
    for (int b = 0; b < total; b += 128) {
        const bool mine = cnt > 0 && start < b + 128 && start + cnt > b;
        const int mpos = start > b ? start - b : 0;
        if (mine) sw.mark[mpos] = lane + 1;
        __builtin_amdgcn_wave_barrier();
        int m0 = sw.mark[lane], m1 = sw.mark[64 + lane];
        __builtin_amdgcn_wave_barrier();
        if (mine) sw.mark[mpos] = 0;
        m0 = bin_scan_max(m0);                       // (slot 0 of an iteration is always marked: m0 >= 1 in every lane)
        m1 = max(bin_scan_max(m1), __builtin_amdgcn_readlane(m0, 63));
        const int4 o0 = sw.own[m0 - 1], o1 = sw.own[m1 - 1];
        uint32_t id0 = 0, id1 = 0;
        if (WITH_ID) { id0 = sw.id[m0 - 1]; id1 = sw.id[m1 - 1]; }
        const int j0 = b + lane, k0 = j0 - o0.x;
        int yy0 = (int)((float)k0 * __builtin_amdgcn_rcpf((float)o0.w));       // k / w: the reciprocal estimate, corrected to the exact quotient
        if (yy0 * o0.w > k0) --yy0;
        else if ((yy0 + 1) * o0.w <= k0) ++yy0;
        const uint32_t t0 = (uint32_t)((o0.z + yy0) * gx + o0.y + (k0 - yy0 * o0.w));
        const bool second = b + 64 < total;
        const int j1 = b + 64 + lane, k1 = j1 - o1.x;
        int yy1 = (int)((float)k1 * __builtin_amdgcn_rcpf((float)o1.w));
        if (yy1 * o1.w > k1) --yy1;
        else if ((yy1 + 1) * o1.w <= k1) ++yy1;
        const uint32_t t1 = (uint32_t)((o1.z + yy1) * gx + o1.y + (k1 - yy1 * o1.w));
        f(j0 < total, t0, id0);
        if (second) f(j1 < total, t1, id1);          // (issuing both steps' counter operations as ONE LDS round trip was measured: no gain)
    }
    __builtin_amdgcn_wave_barrier();                 // (the table is rewritten by the next chunk)
}

__device__ __forceinline__ void bin_unpack(const uint2 r, bool ok, int& cnt, int& minx, int& miny, int& w) {
    minx = (int)(r.x & 0xFFFFu); miny = (int)(r.x >> 16);
    w = (int)(r.y & 0xFFFFu);
    cnt = ok ? w * (int)(r.y >> 16) : 0;
    if (w < 1) w = 1;
}

// A wave keeps ALL of its Gaussians in registers, CH chunks of 64 (loaded once, up front, in one burst): on this part stores and
// loads share one in-order counter (vmcnt), so a load waited for behind the scatter's stores waits for the stores' acknowledgements
// too -- per chunk, with two waves per SIMD to cover it (measured: 57 of the scatter's 82 us).  The chunk loop stays rolled; the
// chunk's registers are picked by a compare-select chain over the (uniform) chunk index.
template <int CH>
__device__ __forceinline__ uint32_t bin_pick(const uint32_t (&a)[CH], int c) {
    uint32_t r = a[0];
    c = __builtin_amdgcn_readfirstlane(c);       // (the chunk index is uniform: the select masks live in scalar registers)
#pragma unroll
    for (int k = 1; k < CH; ++k) {      // (the empty asm keeps the chain a chain: without it the compiler turns it back into an
        r = (c == k) ? a[k] : r;        // indexed load from a scratch copy of the array -- global memory)
        asm volatile("" : "+v"(r));
    }
    return r;
}

// A.  grid = NB blocks of G depth-ordered Gaussians; dynamic LDS: T counters.  Counting is order-free, so the block is 16 waves
// (each walks a 16th of the chunk: the walk is a chain of LDS round trips, and only more waves hide them).
#define BINA_THREADS 1024
#define BINA_WAVES (BINA_THREADS / GP_WAVE)
template <int CH>
__global__ __launch_bounds__(BINA_THREADS) void gp_bin_count_kernel(int N, int gx, int T, int G, const uint2* __restrict__ rect_sorted,
                                                                   uint32_t* __restrict__ hist, uint32_t* __restrict__ wave_cnt,
                                                                   uint32_t* __restrict__ total_slots) {
    extern __shared__ uint32_t s_dyn[];                         // [BIN_WAVES][ceil(T / 2)]: two 16-bit counters per word
    __shared__ BinWave s_w[BINA_WAVES];
    __shared__ uint32_t s_part[BINA_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int words = (T + 1) / 2;
    const int per_wave = G / BINA_WAVES;                        // 16 .. 64 CH Gaussians (small clouds use small blocks)
    const int i_begin = blockIdx.x * G + wave * per_wave;
    int i_end = i_begin + per_wave;
    if (i_end > N) i_end = N;
    uint32_t rx[CH], ry[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = i_begin + c * 64 + lane;
        const uint2 r = rect_sorted[i < N ? i : N - 1];
        rx[c] = r.x; ry[c] = i < i_end ? r.y : 0u;                            // (tiles per row | rows: 0 rows = no instances)
    }
    for (int e = tid; e < BIN_WAVES * words; e += BINA_THREADS) s_dyn[e] = 0u;
    s_w[wave].mark[lane] = 0; s_w[wave].mark[64 + lane] = 0;
    __syncthreads();
    // The scatter kernel's wave q owns the q-th quarter of the chunk = this kernel's waves 4 q .. 4 q + 3: they count into its
    // counters, and the scatter kernel loads them instead of walking the chunk a second time (its counting pass was 13 us).
    uint32_t* q_cnt = s_dyn + (wave / (BINA_WAVES / BIN_WAVES)) * words;
    uint32_t mine_total = 0;
#pragma unroll 1
    for (int c = 0; c < CH; ++c) {
        if (i_begin + c * 64 >= i_end) break;
        int cnt, minx, miny, w;
        bin_unpack(make_uint2(bin_pick<CH>(rx, c), bin_pick<CH>(ry, c)), true, cnt, minx, miny, w);
        mine_total += (uint32_t)cnt;
        bin_walk<false>(lane, s_w[wave], cnt, minx, miny, w, 0u, gx, [&](bool valid, uint32_t tile, uint32_t) {
            if (valid) atomicAdd(&q_cnt[tile >> 1], 1u << (16u * (tile & 1u)));      // (integer LDS atomic, no return: order-free)
        });
    }
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) mine_total += __shfl_xor(mine_total, dd);
    if (lane == 0) s_part[wave] = mine_total;
    __syncthreads();
    uint32_t* wrow = wave_cnt + (size_t)blockIdx.x * BIN_WAVES * words;
    for (int e = tid; e < BIN_WAVES * words; e += BINA_THREADS) wrow[e] = s_dyn[e];
    uint32_t* row = hist + (size_t)blockIdx.x * T;
    for (int i = tid; i < words; i += BINA_THREADS) {             // the block's row: the four waves' counts added up, unpacked
        uint32_t sum = 0;
#pragma unroll
        for (int q = 0; q < BIN_WAVES; ++q) sum += s_dyn[q * words + i];      // (both halves at once: sums stay below 65 536)
        row[2 * i] = sum & 0xFFFFu;
        if (2 * i + 1 < T) row[2 * i + 1] = sum >> 16;
    }
    if (tid == 0) {
        uint32_t tot = 0;
        for (int w2 = 0; w2 < BINA_WAVES; ++w2) tot += s_part[w2];
        if (tot) atomicAdd(total_slots + (blockIdx.x % GP_TOTAL_SLOTS), tot);
    }
}

// B.  grid = ceil(T / 64) workgroups of 16 waves: lane = tile (64 consecutive tiles: every load is one 256-byte line), wave =
// segment of the blocks.  hist[b][t] <- sum over b' < b of hist[b'][t];  totals[t] = the column sum.
__global__ __launch_bounds__(64 * BIN_SEGS) void gp_bin_scan_kernel(uint32_t* __restrict__ hist, int NB, int T, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_tot[BIN_SEGS][64];
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const bool live = t < T;
    const int per = (NB + BIN_SEGS - 1) / BIN_SEGS;
    const int b0 = seg * per;
    int b1 = b0 + per;
    if (b1 > NB) b1 = NB;
    uint32_t sum = 0;
    if (per <= BIN_SEG_MAX) {
        uint32_t v[BIN_SEG_MAX];
#pragma unroll
        for (int k = 0; k < BIN_SEG_MAX; ++k) v[k] = (live && b0 + k < b1) ? hist[(size_t)(b0 + k) * T + t] : 0u;
#pragma unroll
        for (int k = 0; k < BIN_SEG_MAX; ++k) sum += v[k];
        s_tot[seg][lane] = sum;
        __syncthreads();
        uint32_t run = 0, tot = 0;
#pragma unroll
        for (int s2 = 0; s2 < BIN_SEGS; ++s2) { const uint32_t x = s_tot[s2][lane]; run += s2 < seg ? x : 0u; tot += x; }
#pragma unroll
        for (int k = 0; k < BIN_SEG_MAX; ++k) {
            if (live && b0 + k < b1) hist[(size_t)(b0 + k) * T + t] = run;
            run += v[k];
        }
        if (seg == 0 && live) totals[t] = tot;
    } else {                                   // more than 512 blocks: two passes over the column (the second one hits L2)
        for (int b = b0; b < b1; ++b) sum += live ? hist[(size_t)b * T + t] : 0u;
        s_tot[seg][lane] = sum;
        __syncthreads();
        uint32_t run = 0, tot = 0;
        for (int s2 = 0; s2 < BIN_SEGS; ++s2) { const uint32_t x = s_tot[s2][lane]; run += s2 < seg ? x : 0u; tot += x; }
        for (int b = b0; b < b1 && live; ++b) {
            const uint32_t x = hist[(size_t)b * T + t];
            hist[(size_t)b * T + t] = run;
            run += x;
        }
        if (seg == 0 && live) totals[t] = tot;
    }
}

// C.  grid = NB (the blocks of A).  dynamic LDS: s_base[T] u32 | s_cnt[BIN_WAVES][ceil(T / 2)] (two u16 counters per word: a wave's
// quarter of a chunk holds at most G / 4 < 65 536 Gaussians, and a Gaussian puts at most one instance into a tile).
template <int CH>
__global__ __launch_bounds__(BIN_THREADS) void gp_bin_scatter_kernel(int N, int gx, int T,
                                                                     const uint32_t* __restrict__ sorted_ids,
                                                                     const uint2* __restrict__ rect_sorted,
                                                                     const uint32_t* __restrict__ hist_scanned,
                                                                     const uint32_t* __restrict__ totals,
                                                                     const uint32_t* __restrict__ wave_cnt,
                                                                     uint32_t* __restrict__ point_list, uint32_t capacity,
                                                                     int2* __restrict__ ranges, uint32_t* __restrict__ status, int ablate,
                                                                     uint32_t* __restrict__ order, int n_chunks, uint32_t key_tag) {
    extern __shared__ uint32_t s_dyn[];
    __shared__ BinWave s_w[BIN_WAVES];
    __shared__ uint32_t s_wsum[BIN_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == gridDim.x - 1 && order) {
        // One workgroup more than there are chunks: the composite forward's heavy-first launch order (gp_tile_order_kernel's job:
        // a counting sort of the tiles by a logarithm of their list length = the tile totals this kernel reads anyway), computed
        // BESIDE the scatter instead of in a 9 us launch of its own behind it.
        uint32_t* s_cntb = s_dyn;                    // [128] bucket counts, [128] bases, then one bucket byte per tile
        uint32_t* s_baseb = s_dyn + 128;
        uint8_t* s_bk = reinterpret_cast<uint8_t*>(s_dyn + 256);
        if (tid < 128) s_cntb[tid] = 0u;
        __syncthreads();
        for (int t0 = 0; t0 < T; t0 += 16 * BIN_THREADS) {
            uint32_t c[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int t = t0 + k * BIN_THREADS + tid; c[k] = totals[t < T ? t : T - 1]; }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int t = t0 + k * BIN_THREADS + tid;
                if (t < T) { const int b = gp_tile_bucket((int)c[k]); s_bk[t] = (uint8_t)b; atomicAdd(&s_cntb[b], 1u); }
            }
        }
        __syncthreads();
        if (tid < 128) {
            uint32_t run = 0;
#pragma unroll 16
            for (int b = 0; b < 128; ++b) { const uint32_t x = s_cntb[b]; run += b < tid ? x : 0u; }
            s_baseb[tid] = run;
        }
        __syncthreads();
        for (int t = tid; t < T; t += BIN_THREADS) order[atomicAdd(&s_baseb[s_bk[t]], 1u)] = (uint32_t)t;
        return;
    }
    // XCD-aware chunk order: workgroups go to the 8 XCDs round-robin by id, and consecutive slots of a tile's list are written by
    // CONSECUTIVE chunks of the depth-ordered Gaussians (1.5 instances per tile and chunk at configs[2]) -- with chunk = blockIdx they
    // came from eight different L2s, every 4-byte store a partially written 64-byte line of its own on the way to HBM.  Here XCD x
    // owns a contiguous eighth of the chunks: a tile's slots fill up inside ONE L2 and leave it as whole lines.
    const int per_xcd = (n_chunks + 7) / 8;
    const int blk = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || blk >= n_chunks) return;
    constexpr int G = CH * 64 * BIN_WAVES;
    const int i_begin = blk * G + wave * (CH * 64);
    uint32_t rx[CH], ry[CH], rid[CH];           // this wave's Gaussians: tile rectangle and id, the only global loads of the walks
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = i_begin + c * 64 + lane;
        const uint2 r = rect_sorted[i < N ? i : N - 1];
        rx[c] = r.x; ry[c] = i < N ? r.y : 0u;
        rid[c] = sorted_ids[i < N ? i : N - 1];
    }
    const int words = (T + 1) / 2;
    uint32_t* s_base = s_dyn;
    uint32_t* s_cnt = s_dyn + T;
    // ---- tile_start = exclusive scan of the T totals (every block for itself: T <= 8192 values out of L2, ~2 us in parallel with
    // the 500 other blocks, instead of one more launch)
    // (every global load of the prologue leaves up front -- the T totals and this block's row of scanned counts, up to 2 x 32 per
    // thread: as two loops of load -> LDS store they were 2 x 22 dependent round trips to L2, 60 of the kernel's 86 us)
    constexpr int PRE = GP_BIN_MAX_TILES / BIN_THREADS;
    uint32_t tv[PRE], rv[PRE];
    {
        const uint32_t* row = hist_scanned + (size_t)blk * T;
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const int t = tid + k * BIN_THREADS;
            tv[k] = (k * BIN_THREADS < T) ? totals[t < T ? t : T - 1] : 0u;
            rv[k] = (k * BIN_THREADS < T) ? row[t < T ? t : T - 1] : 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
        const int t = tid + k * BIN_THREADS;
        if (t < T) s_base[t] = tv[k];
    }
    {   // the per-(wave, tile) counts of this chunk, counted by gp_bin_count_kernel (coalesced copy)
        const uint32_t* wrow = wave_cnt + (size_t)blk * BIN_WAVES * words;
        const int nw = BIN_WAVES * words;
        for (int i0 = 0; i0 < nw; i0 += 16 * BIN_THREADS) {            // 16 loads in flight per thread, then 16 LDS stores
            uint32_t v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int i = i0 + k * BIN_THREADS + tid; v[k] = wrow[i < nw ? i : nw - 1]; }
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int i = i0 + k * BIN_THREADS + tid; if (i < nw) s_cnt[i] = (ablate & 1) ? 0u : v[k]; }
        }
    }
    s_w[wave].mark[lane] = 0; s_w[wave].mark[64 + lane] = 0;
    __syncthreads();
    {
        const int PT = (T + BIN_THREADS - 1) / BIN_THREADS;
        const int t0 = tid * PT;
        int t1 = t0 + PT;
        if (t1 > T) t1 = T;
        uint32_t loc = 0;
        for (int t = t0; t < t1; ++t) loc += s_base[t];
        uint32_t x = (uint32_t)bin_scan_add((int)loc);
        if (lane == 63) s_wsum[wave] = x;
        __syncthreads();
        uint32_t run = x - loc;
        for (int w2 = 0; w2 < wave; ++w2) run += s_wsum[w2];
        for (int t = t0; t < t1; ++t) {
            const uint32_t c = s_base[t];
            s_base[t] = run;
            run += c;
        }
        if (blk == 0 && tid == BIN_THREADS - 1 && status) {      // (the last thread's running sum is the grand total R)
            status[0] = run;
            status[1] = (run > capacity || (key_tag && status[2] == key_tag)) ? 1u : 0u;      // (key_tag: gp_raster_settings.depth_key_bits)
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
        const int t = tid + k * BIN_THREADS;
        if (t < T) {
            const uint32_t st = s_base[t];
            if (blk == 0) {              // per-tile [start, end) -- cut at the capacity (capacity mode, overflow: flagged above)
                const uint32_t c = tv[k];
                const uint32_t lo = st < capacity ? st : capacity, hi = st + c < capacity ? st + c : capacity;
                ranges[t] = c ? make_int2((int)lo, (int)hi) : make_int2(0, 0);
            }
            s_base[t] = st + rv[k];
        }
    }
    uint32_t* my_cnt = s_cnt + wave * words;
    __syncthreads();
    // ---- exclusive prefix over the waves, both halves of a word at once (sums stay below 65 536)
    for (int i = tid; i < words; i += BIN_THREADS) {
        uint32_t run = 0;
#pragma unroll
        for (int w2 = 0; w2 < BIN_WAVES; ++w2) {
            const uint32_t c = s_cnt[w2 * words + i];
            s_cnt[w2 * words + i] = run;
            run += c;
        }
    }
    __syncthreads();
    // ---- pass 2: the same walk, now placing every instance
    if (!(ablate & 2)) {
#pragma unroll 1
        for (int c = 0; c < CH; ++c) {
            if (i_begin + c * 64 >= N) break;
            int cnt, minx, miny, w;
            bin_unpack(make_uint2(bin_pick<CH>(rx, c), bin_pick<CH>(ry, c)), true, cnt, minx, miny, w);
            const uint32_t id = bin_pick<CH>(rid, c);
            bin_walk<true>(lane, s_w[wave], cnt, minx, miny, w, id, gx, [&](bool valid, uint32_t tile, uint32_t oid) {
                // Rank among the step's lanes that hit the same tile.  Equal tiles inside one step are the exception (the 64
                // instances belong to ~16 Gaussians scattered over the screen), so the step asks the counters instead of matching
                // all lanes against all lanes: read the (wave, tile) count, add one per lane (an atomic that returns nothing), read
                // it again -- three LDS operations issued back to back, executed in program order, ONE round trip.  A count that
                // went up by exactly one had a single hit: its lane's slot is the count before.  Lanes whose count went up by more
                // (0.7 tiles per step at configs[2], in 28 % of the steps) are matched against each other bit by bit (lower lane =
                // earlier instance = in front).  [A `while (tiles left)` loop over those tiles with readlane + ballot measured 3x the
                // cost of this fixed 13-ballot block.]
                const uint32_t sh = 16u * (tile & 1u);
                const uint32_t tsafe = valid ? tile : 0u;
                const uint32_t a_cnt = (uint32_t)(uintptr_t)&my_cnt[tsafe >> 1], a_base = (uint32_t)(uintptr_t)&s_base[tsafe];
                const uint32_t inc = valid ? (1u << sh) : 0u;
                uint32_t w0, w1, tbase;
                // (spelled out: a volatile C++ access to LDS becomes a flat load with its own vmcnt wait)
                asm volatile("ds_read_b32 %0, %3\n\t"
                             "ds_add_u32 %3, %4\n\t"
                             "ds_read_b32 %1, %3\n\t"
                             "ds_read_b32 %2, %5\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(w0), "=&v"(w1), "=&v"(tbase) : "v"(a_cnt), "v"(inc), "v"(a_base) : "memory");
                const uint32_t c0 = (w0 >> sh) & 0xFFFFu, hits = ((w1 >> sh) & 0xFFFFu) - c0;
                uint32_t rank = 0;
                const bool isdup = valid && hits > 1u;
                const unsigned long long dup = __ballot(isdup);
                if (dup) {            // (28 % of the steps at configs[2]) match the duplicated lanes against each other, bit by bit
                    unsigned long long peers = dup;
#pragma unroll
                    for (int b = 0; b < 13; ++b) {               // (T <= GP_BIN_MAX_TILES = 2^13)
                        const bool bit = (tile >> b) & 1u;
                        const unsigned long long mm = __ballot(bit);
                        peers &= bit ? mm : ~mm;
                    }
                    if (isdup) rank = gp_mbcnt(peers);
                }
                if (valid) {
                    const uint32_t pos = tbase + c0 + rank;
                    if (pos < capacity && !(ablate & 4)) point_list[pos] = oid;      // (capacity-mode overflow: the lists are cut at the capacity)
                }
            });
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
// Gaussians per block G = 256 ... 8192 (1 ... 32 register-resident chunks of 64 per wave of the scatter kernel), at most 512
// blocks: up to 4.2 M Gaussians; beyond that -- or beyond GP_BIN_MAX_TILES tiles -- the caller falls back to duplicate + radix sort.
// (gp_debug_option(5, 1) forces that path: the A/B switch of tools/ and of the parity tests)
// The two kernels need the whole per-tile histogram in ONE workgroup's LDS: dynamic (4 + 2 BIN_WAVES) T bytes for the scatter, 2 BIN_WAVES T for
// the count, plus a few KB of static arrays (bounded below by 32 KB: the count kernel's per-wave records).  gfx950 offers 160 KB per
// workgroup; the device's own figure is asked once (round-4 advisor: on a part with a smaller limit the launch would fail where the
// duplicate + radix-sort path could have taken over).  The DPP row_bcast scans of these kernels (gp_wave_scan_add) are gfx9 / CDNA forms:
// the library is built for gfx950 only (__graft_entry__.HIP_FLAGS), nothing else can load it.
static size_t bin_lds_limit() {
    static thread_local int cached_dev = -1;
    static thread_local size_t cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 64 * 1024;
    if (dev != cached_dev) {
        int v = 0;
        cached = (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0) ? (size_t)v : 64 * 1024;
        cached_dev = dev;
    }
    return cached;
}
static size_t bin_lds_need(size_t T) {
    const size_t packed = (size_t)BIN_WAVES * ((T + 1) / 2) * sizeof(uint32_t);
    return 32 * 1024 + T * sizeof(uint32_t) + packed;      // (the larger of the two kernels + the static bound)
}
bool gp_bin_supported(size_t N, size_t T) {
    return N > 0 && N <= 512u * 8192u && T >= 1 && T <= GP_BIN_MAX_TILES && gp_debug_get(5) == 0 && bin_lds_need(T) <= bin_lds_limit();
}

GpBinPlan gp_bin_plan(size_t N, size_t T) {
    GpBinPlan p;
    size_t G = 256;                                       // small clouds: small blocks, so that the grid still spans the chip
    while ((N + G - 1) / G > 512 && G < 8192) G *= 2;     // (10 k Gaussians in blocks of 2048 were FIVE workgroups: 96 us)
    p.G = (int)G;
    p.NB = (int)((N + G - 1) / G);
    p.hist_elems = (size_t)p.NB * T + T + 64 + (size_t)p.NB * BIN_WAVES * ((T + 1) / 2);      // rows + totals + per-wave packed counts
    return p;
}

static inline uint32_t* bin_wave_cnt(const GpBinPlan& p, size_t T, uint32_t* hist) { return hist + (size_t)p.NB * T + T + 64; }

int gp_bin_count(const GpBinPlan& p, size_t N, int gx, size_t T, const uint2* rect_sorted, uint32_t* hist, uint32_t* total_slots, hipStream_t s) {
    const dim3 grid((unsigned)p.NB), block(BINA_THREADS);
    const size_t lds = (size_t)BIN_WAVES * ((T + 1) / 2) * sizeof(uint32_t);
    uint32_t* wave_cnt = bin_wave_cnt(p, T, hist);
    if (lds > 48 * 1024) {
        static thread_local size_t lds_set[4] = {0, 0, 0, 0};
        const int v = p.G <= 1024 ? 0 : p.G == 2048 ? 1 : p.G == 4096 ? 2 : 3;
        if (lds > lds_set[v]) {
            const void* f = v == 0 ? (const void*)gp_bin_count_kernel<1> : v == 1 ? (const void*)gp_bin_count_kernel<2> : v == 2 ? (const void*)gp_bin_count_kernel<4>
                                                                                                                                   : (const void*)gp_bin_count_kernel<8>;
            GP_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            lds_set[v] = lds;
        }
    }
    if (p.G <= 1024) hipLaunchKernelGGL(gp_bin_count_kernel<1>, grid, block, lds, s, (int)N, gx, (int)T, p.G, rect_sorted, hist, wave_cnt, total_slots);
    else if (p.G == 2048) hipLaunchKernelGGL(gp_bin_count_kernel<2>, grid, block, lds, s, (int)N, gx, (int)T, p.G, rect_sorted, hist, wave_cnt, total_slots);
    else if (p.G == 4096) hipLaunchKernelGGL(gp_bin_count_kernel<4>, grid, block, lds, s, (int)N, gx, (int)T, p.G, rect_sorted, hist, wave_cnt, total_slots);
    else if (p.G == 8192) hipLaunchKernelGGL(gp_bin_count_kernel<8>, grid, block, lds, s, (int)N, gx, (int)T, p.G, rect_sorted, hist, wave_cnt, total_slots);
    else GP_FAIL("bin: unsupported block size %d", p.G);
    GP_LAUNCH_CHECK();
    return 0;
}

template <int CH>
static int bin_scatter_launch(const GpBinPlan& p, size_t N, int gx, size_t T, const uint32_t* sorted_ids, const uint2* rect_sorted, const uint32_t* hist,
                              const uint32_t* totals, const uint32_t* wave_cnt, uint32_t* point_list, uint32_t capacity, int2* ranges, uint32_t* status, size_t lds,
                              uint32_t* order, uint32_t key_tag, hipStream_t s) {
    static thread_local size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        GP_HIP_CHECK(hipFuncSetAttribute((const void*)gp_bin_scatter_kernel<CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL(gp_bin_scatter_kernel<CH>, dim3(8u * (((unsigned)p.NB + 7u) / 8u) + (order ? 1u : 0u)), dim3(BIN_THREADS), lds, s, (int)N, gx, (int)T, sorted_ids,
                       rect_sorted, hist, totals, wave_cnt, point_list, capacity, ranges, status, gp_debug_get(6), order, p.NB, key_tag);
    GP_LAUNCH_CHECK();
    return 0;
}

int gp_bin_scatter(const GpBinPlan& p, size_t N, int gx, size_t T, const uint32_t* sorted_ids, const uint2* rect_sorted, uint32_t* hist,
                   uint32_t* point_list, uint32_t capacity, int2* ranges, uint32_t* status, uint32_t* order, uint32_t key_tag, hipStream_t s) {
    uint32_t* totals = hist + (size_t)p.NB * T;
    {
        GpProfScope _p("bin_scan", s);
        hipLaunchKernelGGL(gp_bin_scan_kernel, dim3((unsigned)((T + 63) / 64)), dim3(64 * BIN_SEGS), 0, s, hist, p.NB, (int)T, totals);
        GP_LAUNCH_CHECK();
    }
    size_t lds = (T + BIN_WAVES * ((T + 1) / 2)) * sizeof(uint32_t);
    if (order && lds < 1024 + T + 4) lds = 1024 + T + 4;        // (the tile-order workgroup's two 128-entry tables and a byte per tile)
    if (gp_debug_get(6)) GP_HIP_CHECK(hipMemsetAsync(point_list, 0, (size_t)capacity * 4, s));   // (ablation runs leave slots unwritten: id 0 is a valid one)
    GpProfScope _p("bin_scatter", s);
    if (p.G == 256) return bin_scatter_launch<1>(p, N, gx, T, sorted_ids, rect_sorted, hist, totals, bin_wave_cnt(p, T, hist), point_list, capacity, ranges, status, lds, order, key_tag, s);
    if (p.G == 512) return bin_scatter_launch<2>(p, N, gx, T, sorted_ids, rect_sorted, hist, totals, bin_wave_cnt(p, T, hist), point_list, capacity, ranges, status, lds, order, key_tag, s);
    if (p.G == 1024) return bin_scatter_launch<4>(p, N, gx, T, sorted_ids, rect_sorted, hist, totals, bin_wave_cnt(p, T, hist), point_list, capacity, ranges, status, lds, order, key_tag, s);
    if (p.G == 2048) return bin_scatter_launch<8>(p, N, gx, T, sorted_ids, rect_sorted, hist, totals, bin_wave_cnt(p, T, hist), point_list, capacity, ranges, status, lds, order, key_tag, s);
    if (p.G == 4096) return bin_scatter_launch<16>(p, N, gx, T, sorted_ids, rect_sorted, hist, totals, bin_wave_cnt(p, T, hist), point_list, capacity, ranges, status, lds, order, key_tag, s);
    if (p.G == 8192) return bin_scatter_launch<32>(p, N, gx, T, sorted_ids, rect_sorted, hist, totals, bin_wave_cnt(p, T, hist), point_list, capacity, ranges, status, lds, order, key_tag, s);
    GP_FAIL("bin: unsupported block size %d", p.G);
}

// =====================================================================================================================================
// Depth sort of the N Gaussians with the same three-kernel counting pass, 11 bits at a time.
//
// Rounds 1-3 sorted the 32-bit depth keys in four 8-bit LSD passes of three launches each (block histograms, one row scan per
// digit, scatter): twelve launches of 5 - 13 us for a million keys, every one at its launch / latency floor.  With 2048-bin
// histograms in LDS (8 KB) the key has three digits (11 + 11 + 10 bits), and a pass is the binning stage's own shape:
//   count    block b (KPB = 2048 ... 8192 keys) counts its keys per digit in LDS and writes the row hist[b][0..2048)
//   scan     gp_bin_scan_kernel, as for the tiles: per digit the exclusive prefix over the blocks + the digit totals
//   scatter  digit_start = exclusive scan of the totals (every block for itself), per-(wave, digit) 16-bit counters for the order
//            across the block's four waves, ranks inside a 64-key step from the counters (read, add, read) and masked match-any
// Stable for the same reason the binning is: blocks, waves, steps and lanes are walked in input order.  The last pass carries every
// Gaussian's tile rectangle into depth order (GpSortEpilogue), as the radix sort's last pass did.
// =====================================================================================================================================
#define DS_BITS 11
#define DS_BINS (1 << DS_BITS)
#define DS_THREADS 256
#define DS_WAVES (DS_THREADS / GP_WAVE)

template <int IT>      // keys per thread; a block owns IT * 256 consecutive keys
__global__ __launch_bounds__(DS_THREADS) void gp_dsort_count_kernel(const uint32_t* __restrict__ keys, int n, int shift, uint32_t mask,
                                                                    uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_hist[DS_BINS];
    const int tid = threadIdx.x;
    const int nb_all = (n + IT * DS_THREADS - 1) / (IT * DS_THREADS), per_xcd = (nb_all + 7) / 8;
    const int blk = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);      // XCD x owns a contiguous eighth of the blocks (see gp_bin_scatter_kernel)
    if ((int)(blockIdx.x >> 3) >= per_xcd || blk >= nb_all) return;
    const int base = blk * (IT * DS_THREADS);
    uint32_t k[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = base + it * DS_THREADS + tid;
        k[it] = keys[idx < n ? idx : n - 1];
    }
    for (int d = tid; d < DS_BINS; d += DS_THREADS) s_hist[d] = 0u;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it)
        if (base + it * DS_THREADS + tid < n) atomicAdd(&s_hist[(k[it] >> shift) & mask], 1u);
    __syncthreads();
    uint32_t* row = hist + (size_t)blk * DS_BINS;
    for (int d = tid; d < DS_BINS; d += DS_THREADS) row[d] = s_hist[d];
}

template <int IT>
__global__ __launch_bounds__(DS_THREADS) void gp_dsort_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                      uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                      const uint32_t* __restrict__ hist_scanned,
                                                                      const uint32_t* __restrict__ totals, int n, int shift, uint32_t mask,
                                                                      int nbits, GpSortEpilogue ep) {
    __shared__ uint32_t s_base[DS_BINS];
    __shared__ uint32_t s_cnt[DS_WAVES][DS_BINS / 2];          // two 16-bit counters per word (a wave holds IT * 64 <= 2048 keys)
    __shared__ uint32_t s_wsum[DS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER_WAVE = IT * 64;
    const int nb_all = (n + IT * DS_THREADS - 1) / (IT * DS_THREADS), per_xcd = (nb_all + 7) / 8;
    const int blk = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);      // XCD x owns a contiguous eighth of the blocks
    if ((int)(blockIdx.x >> 3) >= per_xcd || blk >= nb_all) return;
    const int wbase = blk * (IT * DS_THREADS) + wave * PER_WAVE;       // wave w owns keys [w * PER_WAVE, (w + 1) * PER_WAVE)
    uint32_t k[IT], v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int idx = wbase + it * 64 + lane;
        const int ic = idx < n ? idx : n - 1;
        k[it] = keys_in[ic];
        v[it] = vals_in ? vals_in[ic] : (uint32_t)idx;         // vals_in == NULL: the values are the indices (first pass)
    }
    constexpr int PB = DS_BINS / DS_THREADS;                  // 8 digits per thread
    uint32_t tv[PB], rv[PB];
    {
        const uint32_t* row = hist_scanned + (size_t)blk * DS_BINS;
#pragma unroll
        for (int j = 0; j < PB; ++j) { tv[j] = totals[tid * PB + j]; rv[j] = row[tid * PB + j]; }
    }
    for (int i = tid; i < DS_WAVES * (DS_BINS / 2); i += DS_THREADS) (&s_cnt[0][0])[i] = 0u;
    // digit_start = exclusive scan of the totals: thread t owns digits [8 t, 8 t + 8)
    {
        uint32_t loc = 0;
#pragma unroll
        for (int j = 0; j < PB; ++j) loc += tv[j];
        const uint32_t x = (uint32_t)bin_scan_add((int)loc);
        if (lane == 63) s_wsum[wave] = x;
        __syncthreads();
        uint32_t run = x - loc;
        for (int w2 = 0; w2 < wave; ++w2) run += s_wsum[w2];
#pragma unroll
        for (int j = 0; j < PB; ++j) { s_base[tid * PB + j] = run + rv[j]; run += tv[j]; }
    }
    __syncthreads();
    uint32_t* my_cnt = s_cnt[wave];
    // pass 1: this wave's keys per digit
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const uint32_t d = (k[it] >> shift) & mask;
        if (wbase + it * 64 + lane < n) atomicAdd(&my_cnt[d >> 1], 1u << (16u * (d & 1u)));
    }
    __syncthreads();
    for (int i = tid; i < DS_BINS / 2; i += DS_THREADS) {      // exclusive prefix over the waves, both halves of a word at once
        uint32_t run = 0;
#pragma unroll
        for (int w2 = 0; w2 < DS_WAVES; ++w2) { const uint32_t c = s_cnt[w2][i]; s_cnt[w2][i] = run; run += c; }
    }
    __syncthreads();
    // pass 2: place.  Rank inside a step: the digit's counter before / after one add per lane, then the lanes that share a digit
    // (most steps have some: 64 keys into 2048 bins, and the top digit takes a handful of values) are matched bit by bit.
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const bool valid = wbase + it * 64 + lane < n;
        const uint32_t d = (k[it] >> shift) & mask;
        const uint32_t sh = 16u * (d & 1u);
        const uint32_t a_cnt = (uint32_t)(uintptr_t)&my_cnt[d >> 1], a_base = (uint32_t)(uintptr_t)&s_base[d];
        const uint32_t inc = valid ? (1u << sh) : 0u;
        uint32_t w0, w1, dbase;
        asm volatile("ds_read_b32 %0, %3\n\t"
                     "ds_add_u32 %3, %4\n\t"
                     "ds_read_b32 %1, %3\n\t"
                     "ds_read_b32 %2, %5\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(w0), "=&v"(w1), "=&v"(dbase) : "v"(a_cnt), "v"(inc), "v"(a_base) : "memory");
        const uint32_t c0 = (w0 >> sh) & 0xFFFFu, hits = ((w1 >> sh) & 0xFFFFu) - c0;
        uint32_t rank = 0;
        const bool isdup = valid && hits > 1u;
        const unsigned long long dup = __ballot(isdup);
        if (dup) {
            unsigned long long peers = dup;
            for (int b = 0; b < nbits; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long mm = __ballot(bit);
                peers &= bit ? mm : ~mm;
            }
            if (isdup) rank = gp_mbcnt(peers);
        }
        if (valid) {
            const uint32_t pos = dbase + c0 + rank;
            keys_out[pos] = k[it];
            vals_out[pos] = v[it];
            if (ep.by_value) {          // last pass: the Gaussian's tile rectangle (and tile count) follows its id into depth order
                const uint2 rr = ep.by_value[v[it]];
                ep.sorted_out[pos] = rr;
                ep.count_out[pos] = (rr.y & 0xFFFFu) * (rr.y >> 16);
            }
        }
    }
}

// MEASURED SLOWER than the four 8-bit radix passes at configs[2] and therefore OFF by default (gp_debug_option(8, 2) selects it; the
// parity tests run both): 3 x (count 5.8 + scan 5.6 + scatter 25.4 us) = 0.113 ms against 4 x (5.8 + 5.0 + 12.6) = 0.089 ms
// (profiles/r04_depth_sort_ab.txt).  Nine launches instead of twelve do not pay for a scatter that is twice as long: with 2048 bins
// and 2048 keys per block every key is its own digit run (no coalescing to stage for), and the kernel is one latency chain -- key
// loads, a 2048-entry prefix, eight counter round trips per wave -- at two workgroups per CU.
// (Round 6: with the XCD-aware block order of both sorts -- profiles/r06_xcd_locality_ab.txt -- 0.090 ms against 0.080 for the four radix passes: still off.)
bool gp_dsort_supported(size_t n) { return n > 0 && n <= 512u * 8192u && gp_debug_get(8) == 2; }
size_t gp_dsort_hist_elems(size_t n) {
    size_t kpb = 2048;
    while ((n + kpb - 1) / kpb > 512 && kpb < 8192) kpb *= 2;
    return ((n + kpb - 1) / kpb) * DS_BINS + DS_BINS + 64;
}

template <int IT>
static int dsort_pass(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, uint32_t* hist, int n, int nb, int shift, int bits,
                      const GpSortEpilogue& ep, hipStream_t s) {
    const uint32_t mask = (1u << bits) - 1u;
    uint32_t* totals = hist + (size_t)nb * DS_BINS;
    hipLaunchKernelGGL(gp_dsort_count_kernel<IT>, dim3(8u * (((unsigned)nb + 7u) / 8u)), dim3(DS_THREADS), 0, s, kin, n, shift, mask, hist);
    GP_LAUNCH_CHECK();
    hipLaunchKernelGGL(gp_bin_scan_kernel, dim3(DS_BINS / 64), dim3(64 * BIN_SEGS), 0, s, hist, nb, DS_BINS, totals);
    GP_LAUNCH_CHECK();
    hipLaunchKernelGGL(gp_dsort_scatter_kernel<IT>, dim3(8u * (((unsigned)nb + 7u) / 8u)), dim3(DS_THREADS), 0, s, kin, vin, kout, vout, (const uint32_t*)hist,
                       (const uint32_t*)totals, n, shift, mask, bits, ep);
    GP_LAUNCH_CHECK();
    return 0;
}

// keys in b.k[0] (values = indices); returns the index of the buffer pair holding the result (1: three passes), or -1
int gp_depth_sort3(GpSortBufs& b, size_t n, uint32_t* hist, hipStream_t s, const GpSortEpilogue* epilogue) {
    if (n == 0) return 0;
    size_t kpb = 2048;
    while ((n + kpb - 1) / kpb > 512 && kpb < 8192) kpb *= 2;
    const int nb = (int)((n + kpb - 1) / kpb);
    const int shifts[3] = {0, 11, 22}, bits[3] = {11, 11, 10};
    int cur = 0;
    for (int p = 0; p < 3; ++p) {
        GpSortEpilogue ep = {nullptr, nullptr, nullptr};
        if (epilogue && p == 2) ep = *epilogue;
        const uint32_t* vin = p == 0 ? nullptr : b.v[cur];
        int rc;
        if (kpb == 2048) rc = dsort_pass<8>(b.k[cur], vin, b.k[cur ^ 1], b.v[cur ^ 1], hist, (int)n, nb, shifts[p], bits[p], ep, s);
        else if (kpb == 4096) rc = dsort_pass<16>(b.k[cur], vin, b.k[cur ^ 1], b.v[cur ^ 1], hist, (int)n, nb, shifts[p], bits[p], ep, s);
        else rc = dsort_pass<32>(b.k[cur], vin, b.k[cur ^ 1], b.v[cur ^ 1], hist, (int)n, nb, shifts[p], bits[p], ep, s);
        if (rc) return -1;
        cur ^= 1;
    }
    return cur;
}
