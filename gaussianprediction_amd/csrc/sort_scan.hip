// sort_scan.hip -- device-wide exclusive scan and STABLE LSD radix sort of (u32 key, u32 value) pairs,
// hand-written for wave64 / gfx950.  Used by the tile binning step of the rasterizer:
//   (1) sort the N Gaussians by view depth (32 key bits), (2) emit one (tile, id) pair per touched
//   tile in that depth order, (3) stable-sort the R pairs by tile id (ceil(log2 T) key bits).
// The result is identical to one stable 64-bit sort of (tile<<32 | depth_bits) with Gaussian-id
// tie-break -- the ordering the public 3DGS rasterizer obtains from a stable device radix sort --
// at 4N + 2R element-passes instead of 6R  (stability is what "tile
// ordering bit-exact" requires).
#include "gp_common.h"

// ------------------------------------------------------------------------------------------------
// exclusive scan (u32) inside blocks of 2048, in place (the consumer finishes it: gp_scan_blocks_u32)
// ------------------------------------------------------------------------------------------------
#define SCAN_ITEMS 8
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)
static_assert(SCAN_TILE == GP_SCAN_TILE, "gp_duplicate_kernel finishes the scan of blocks of GP_SCAN_TILE elements");

__global__ __launch_bounds__(SCAN_BLOCK) void gp_scan_block_kernel(uint32_t* __restrict__ data, size_t n,
                                                                  uint32_t* __restrict__ block_sums, uint32_t* __restrict__ total) {
    __shared__ uint32_t s_wave[SCAN_BLOCK / GP_WAVE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)tid * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t tsum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? data[base + i] : 0u;
        tsum += v[i];
    }
    uint32_t x = tsum;
        x = (uint32_t)gp_wave_scan_add((int)x);
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += s_wave[w];
    uint32_t run = wave_off + x - tsum;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) data[base + i] = run;
        run += v[i];
    }
    if (tid == SCAN_BLOCK - 1) {
        block_sums[blockIdx.x] = wave_off + x;
        if (total) atomicAdd(total + (blockIdx.x % GP_TOTAL_SLOTS), wave_off + x);      // (integer: the sum does not depend on the order)
    }
}

// One-launch form for consumers that can finish the scan themselves (gp_duplicate_kernel): block-local exclusive scan in place,
// block_sums[b] = the block's total, total[0 .. GP_TOTAL_SLOTS) += all of it, spread over the slots (the caller zeroes them).  The consumer adds the sum of the block sums
// in front of its block -- a few hundred values it reduces in one step -- instead of two more launches doing that for it.
int gp_scan_blocks_u32(uint32_t* data, size_t n, uint32_t* block_sums, uint32_t* total, hipStream_t s) {
    if (n == 0) return 0;
    const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(gp_scan_block_kernel, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, s, data, n, block_sums, total);
    GP_LAUNCH_CHECK();
    return 0;
}

size_t gp_scan_tmp_elems(size_t n) {
    size_t total = 256 + 64;   // the radix sort keeps its 256 digit totals here
    while (n > 1) {
        size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
        total += gp_align_up(nb, 64);
        if (nb == 1) break;
        n = nb;
    }
    return total;
}

// ------------------------------------------------------------------------------------------------
// radix sort, 8-bit digits, 256 threads x 16 items per block
// ------------------------------------------------------------------------------------------------
#define RS_BLOCK 256
// keys per workgroup = ITEMS * 256.  16 items amortise the per-block prologue on large inputs (the R tile-splat instances); the
// depth sort of ~1M Gaussians is fastest at 8 (measured at N = 1M: 4 items 0.094 ms, 8 items 0.088 ms, 16 items 0.104 ms -- 245
// workgroups no longer cover the 256 CUs; the tile sort of R = 4.1M pairs: 16 items 0.1015 ms, 8 items 0.098 ms); small inputs use 4
// so that the grid still spans the chip.
#define RS_ITEMS_LARGE 16
#define RS_ITEMS_MID 8
#define RS_ITEMS_SMALL 4
#define RS_SMALL_N (256u << 10)
#define RS_MID_N (8u << 20)
static inline int rs_items_for(size_t n) { return n <= RS_SMALL_N ? RS_ITEMS_SMALL : (n <= RS_MID_N ? RS_ITEMS_MID : RS_ITEMS_LARGE); }

size_t gp_sort_hist_elems(size_t n) {
    const size_t tile = (size_t)rs_items_for(n) * RS_BLOCK;
    return 256 * ((n + tile - 1) / tile) + 64;
}

// XCD-aware block order (round 6): workgroups go to the 8 XCDs round-robin by id; with block = blockIdx the adjacent runs of one digit
// (32 bytes per block at 2048 keys and 256 digits) were written from different L2s, every 64-byte line twice and partially.  XCD x owns
// a contiguous eighth of the blocks: the launches below use 8 ceil(nblocks / 8) workgroups, ids beyond nblocks leave at once.
__device__ __forceinline__ uint32_t rs_logical_block(uint32_t b, uint32_t nblocks) {
    const uint32_t per = (nblocks + 7u) / 8u;
    return (b >> 3) < per ? (b & 7u) * per + (b >> 3) : nblocks;
}
static inline unsigned rs_grid(uint32_t nblocks) { return 8u * ((nblocks + 7u) / 8u); }
// per-block digit histogram, written digit-major: hist[digit * nblocks + block]
template <int RS_ITEMS>
__global__ __launch_bounds__(RS_BLOCK) void gp_radix_hist_kernel(const uint32_t* __restrict__ keys, size_t n, int shift,
                                                                 uint32_t mask, uint32_t* __restrict__ hist,
                                                                 uint32_t nblocks) {
    constexpr int RS_TILE = RS_ITEMS * RS_BLOCK;
    __shared__ uint32_t s_hist[256];
    const int tid = threadIdx.x;
    const uint32_t blk = rs_logical_block(blockIdx.x, nblocks);
    if (blk >= nblocks) return;
    s_hist[tid] = 0;
    __syncthreads();
    const size_t base = (size_t)blk * RS_TILE;
    uint32_t k[RS_ITEMS];
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {        // every load in flight before the first use (as `if (idx < n) atomicAdd(.. keys[idx] ..)`
        const size_t idx = base + (size_t)it * RS_BLOCK + tid;                     // each load was waited for on its own)
        k[it] = keys[idx < n ? idx : n - 1];
    }
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it)
        if (base + (size_t)it * RS_BLOCK + tid < n) atomicAdd(&s_hist[(k[it] >> shift) & mask], 1u);
    __syncthreads();
    hist[(size_t)tid * nblocks + blk] = s_hist[tid];
}

// exclusive scan of every digit's row  hist[digit][0..nblocks)  in place (one workgroup per digit) + the row totals.
// Together with a 256-entry scan of the totals inside the scatter kernel this replaces a generic three-launch device
// scan of the whole 256 x nblocks array: 3 launches per radix pass instead of 5.
__global__ __launch_bounds__(256) void gp_radix_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t nblocks,
                                                               uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    // up to RSC chunks of 256 counters are fetched up front (one round trip instead of one per chunk: the tile sort's rows are
    // 2 016 counters long, eight dependent loads before); longer rows fall through to the chunk-at-a-time loop
    constexpr int RSC = 16;
    uint32_t pre[RSC];
    const bool batched = nblocks <= 256u * RSC;
    if (batched) {
#pragma unroll
        for (int c = 0; c < RSC; ++c) {
            const uint32_t b = 256u * c + tid;
            pre[c] = (256u * c < nblocks) ? row[b < nblocks ? b : nblocks - 1] : 0u;
        }
    }
    int c = 0;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 256, ++c) {
        const uint32_t b = b0 + tid;
        uint32_t v;
        if (batched) {
            v = 0u;
#pragma unroll
            for (int k = 0; k < RSC; ++k) v = (k == c) ? pre[k] : v;       // (register select: no dynamic indexing)
            v = b < nblocks ? v : 0u;
        } else {
            v = b < nblocks ? row[b] : 0u;
        }
        uint32_t x = v;
        x = (uint32_t)gp_wave_scan_add((int)x);
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        uint32_t off = s_carry;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (b < nblocks) row[b] = off + x - v;
        __syncthreads();
        if (tid == 255) s_carry = off + x;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = s_carry;
}

// stable scatter.  Element order inside a block: (wave, iteration, lane); wave w owns the
// contiguous sub-chunk [w*1024, (w+1)*1024) of the block's 4096 elements.
template <int RS_ITEMS>
__global__ __launch_bounds__(RS_BLOCK) void gp_radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                    const uint32_t* __restrict__ vals_in,
                                                                    uint32_t* __restrict__ keys_out,
                                                                    uint32_t* __restrict__ vals_out,
                                                                    const uint32_t* __restrict__ hist_scanned,
                                                                    const uint32_t* __restrict__ totals, size_t n,
                                                                    int shift, uint32_t mask, uint32_t nblocks, GpSortEpilogue ep) {
    constexpr int RS_TILE = RS_ITEMS * RS_BLOCK;
    __shared__ uint32_t s_cnt[RS_BLOCK / GP_WAVE][256];
    __shared__ uint32_t s_dbase[256], s_dw[4];
    __shared__ uint32_t s_k[RS_TILE], s_v[RS_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t blk = rs_logical_block(blockIdx.x, nblocks);
    if (blk >= nblocks) return;
#pragma unroll
    for (int w = 0; w < RS_BLOCK / GP_WAVE; ++w) s_cnt[w][tid] = 0;
    __syncthreads();
    const size_t base = (size_t)blk * RS_TILE + (size_t)wave * (GP_WAVE * RS_ITEMS);
    uint32_t k[RS_ITEMS], v[RS_ITEMS], r[RS_ITEMS];
    // every global load of the block up front (clamped addresses): inside the ranking loop each pair was waited for before its
    // eight ballots -- RS_ITEMS dependent round trips per block, plus one each for the two scan inputs below
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const size_t idx = base + (size_t)it * GP_WAVE + lane;
        const size_t ic = idx < n ? idx : n - 1;
        k[it] = keys_in[ic];
        v[it] = vals_in ? vals_in[ic] : (uint32_t)idx;                       // vals_in == NULL: the values are the indices (first pass)
    }
    const uint32_t tv_pre = totals[tid];
    const uint32_t hs_pre = hist_scanned[(size_t)tid * nblocks + blk];
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const size_t idx = base + (size_t)it * GP_WAVE + lane;
        const bool valid = idx < n;
        if (!valid) { k[it] = 0xFFFFFFFFu; v[it] = 0u; }
        const uint32_t digit = (k[it] >> shift) & mask;
        // peers = lanes of this wave (valid only) holding the same digit
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (digit >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t below = gp_mbcnt(peers);
        const uint32_t cnt = (uint32_t)__popcll(peers);
        uint32_t old = 0;
        if (valid && below == 0) {  // leader of its digit group
            old = s_cnt[wave][digit];
            s_cnt[wave][digit] = old + cnt;
        }
        const int leader = peers ? (__ffsll((long long)peers) - 1) : lane;
        old = __shfl(old, leader);
        r[it] = old + below;
    }
    __syncthreads();
    {   // global base of digit `tid` = exclusive scan of the 256 row totals
        const uint32_t tv = tv_pre;
        uint32_t x = tv;
        x = (uint32_t)gp_wave_scan_add((int)x);
        if (lane == 63) s_dw[wave] = x;
        __syncthreads();
        uint32_t off = 0;
        for (int w = 0; w < wave; ++w) off += s_dw[w];
        s_dbase[tid] = off + x - tv;
    }
    {   // per-digit: exclusive prefix over the 4 waves (block-local), block-local digit base, global base of (digit, block)
        uint32_t c[RS_BLOCK / GP_WAVE], bc = 0;
#pragma unroll
        for (int w = 0; w < RS_BLOCK / GP_WAVE; ++w) { c[w] = s_cnt[w][tid]; bc += c[w]; }
        // block-local exclusive scan of the 256 digit counts
        uint32_t x = bc;
        x = (uint32_t)gp_wave_scan_add((int)x);
        __syncthreads();                 // s_dw is reused
        if (lane == 63) s_dw[wave] = x;
        __syncthreads();
        uint32_t off = 0;
        for (int w = 0; w < wave; ++w) off += s_dw[w];
        const uint32_t lbase = off + x - bc;
        uint32_t run = lbase;
#pragma unroll
        for (int w = 0; w < RS_BLOCK / GP_WAVE; ++w) { s_cnt[w][tid] = run; run += c[w]; }
        // element at block-local sorted position p with this digit goes to global  p + s_dbase[digit]
        s_dbase[tid] = s_dbase[tid] + hs_pre - lbase;
    }
    __syncthreads();
    // stage the block's pairs in sorted order in LDS, then write them out: consecutive threads -> consecutive addresses
    // inside each digit run (instead of every lane scattering 4 bytes on its own)
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const size_t idx = base + (size_t)it * GP_WAVE + lane;
        if (idx < n) {
            const uint32_t digit = (k[it] >> shift) & mask;
            const uint32_t lp = s_cnt[wave][digit] + r[it];
            s_k[lp] = k[it];
            s_v[lp] = v[it];
        }
    }
    __syncthreads();
    const size_t bbase = (size_t)blk * RS_TILE;
    const uint32_t bn = (uint32_t)(n - bbase < RS_TILE ? n - bbase : RS_TILE);
    if (!ep.by_value) {
        for (uint32_t p = tid; p < bn; p += RS_BLOCK) {
            const uint32_t kk = s_k[p];
            const uint32_t pos = p + s_dbase[(kk >> shift) & mask];
            keys_out[pos] = kk;
            vals_out[pos] = s_v[p];
        }
    } else {                    // last pass of the depth sort: the per-Gaussian tile rectangle follows its id into sorted order
        uint32_t pos[RS_ITEMS], vv[RS_ITEMS];
        uint2 rr[RS_ITEMS];
#pragma unroll
        for (int it = 0; it < RS_ITEMS; ++it) {                 // (the gathers of all of a thread's elements in flight together)
            const uint32_t p = tid + it * RS_BLOCK;
            const bool ok = p < bn;
            const uint32_t kk = ok ? s_k[p] : 0u;
            vv[it] = ok ? s_v[p] : 0u;
            pos[it] = p + s_dbase[(kk >> shift) & mask];
            if (ok) { keys_out[pos[it]] = kk; vals_out[pos[it]] = vv[it]; }
            rr[it] = ep.by_value[vv[it]];
        }
#pragma unroll
        for (int it = 0; it < RS_ITEMS; ++it) {
            if (tid + it * RS_BLOCK < bn) {
                ep.sorted_out[pos[it]] = rr[it];
                ep.count_out[pos[it]] = (rr[it].y & 0xFFFFu) * (rr[it].y >> 16);
            }
        }
    }
}

// the histogram kernel above partitions by block only (order inside a block is irrelevant for
// counts), the scatter kernel uses the same block partition [b*4096, (b+1)*4096).
int gp_radix_sort_pairs(GpSortBufs& b, size_t n, int nbits, hipStream_t s, bool iota_vals, const GpSortEpilogue* epilogue) {
    if (n == 0 || nbits <= 0) return 0;
    const int items = rs_items_for(n);
    const size_t tile = (size_t)items * RS_BLOCK;
    const uint32_t nblocks = (uint32_t)((n + tile - 1) / tile);
    int cur = 0;
    // the passes share the key bits evenly (13-bit tile ids: 7 + 6 rather than 8 + 5 -- the first pass's scatter then writes runs of
    // 16 keys = 64 bytes per digit and block instead of 8; 32-bit depth keys: 8 + 8 + 8 + 8 as before)
    const int npass = (nbits + 7) / 8, per = (nbits + npass - 1) / npass;
    for (int shift = 0; shift < nbits; shift += per) {
        const int bits = (nbits - shift) < per ? (nbits - shift) : per;
        const uint32_t mask = (1u << bits) - 1u;
        const uint32_t* vin = (shift == 0 && iota_vals) ? nullptr : b.v[cur];
        GpSortEpilogue ep = {nullptr, nullptr, nullptr};
        if (epilogue && shift + per >= nbits) ep = *epilogue;      // (last pass)
        if (items == RS_ITEMS_SMALL)
            hipLaunchKernelGGL((gp_radix_hist_kernel<RS_ITEMS_SMALL>), dim3(rs_grid(nblocks)), dim3(RS_BLOCK), 0, s, b.k[cur], n, shift, mask,
                               b.hist, nblocks);
        else if (items == RS_ITEMS_MID)
            hipLaunchKernelGGL((gp_radix_hist_kernel<RS_ITEMS_MID>), dim3(rs_grid(nblocks)), dim3(RS_BLOCK), 0, s, b.k[cur], n, shift, mask,
                               b.hist, nblocks);
        else
            hipLaunchKernelGGL((gp_radix_hist_kernel<RS_ITEMS_LARGE>), dim3(rs_grid(nblocks)), dim3(RS_BLOCK), 0, s, b.k[cur], n, shift, mask,
                               b.hist, nblocks);
        if (hipGetLastError() != hipSuccess) { snprintf(gp_err_buf, sizeof(gp_err_buf), "radix hist launch failed"); return -1; }
        if (b.scan_tmp_elems < 256) { snprintf(gp_err_buf, sizeof(gp_err_buf), "radix sort: temp storage too small"); return -1; }
        hipLaunchKernelGGL(gp_radix_rowscan_kernel, dim3(256), dim3(256), 0, s, b.hist, nblocks, b.scan_tmp);
        if (items == RS_ITEMS_SMALL)
            hipLaunchKernelGGL((gp_radix_scatter_kernel<RS_ITEMS_SMALL>), dim3(rs_grid(nblocks)), dim3(RS_BLOCK), 0, s, b.k[cur], vin,
                               b.k[cur ^ 1], b.v[cur ^ 1], b.hist, b.scan_tmp, n, shift, mask, nblocks, ep);
        else if (items == RS_ITEMS_MID)
            hipLaunchKernelGGL((gp_radix_scatter_kernel<RS_ITEMS_MID>), dim3(rs_grid(nblocks)), dim3(RS_BLOCK), 0, s, b.k[cur], vin,
                               b.k[cur ^ 1], b.v[cur ^ 1], b.hist, b.scan_tmp, n, shift, mask, nblocks, ep);
        else
            hipLaunchKernelGGL((gp_radix_scatter_kernel<RS_ITEMS_LARGE>), dim3(rs_grid(nblocks)), dim3(RS_BLOCK), 0, s, b.k[cur], vin,
                               b.k[cur ^ 1], b.v[cur ^ 1], b.hist, b.scan_tmp, n, shift, mask, nblocks, ep);
        if (hipGetLastError() != hipSuccess) { snprintf(gp_err_buf, sizeof(gp_err_buf), "radix scatter launch failed"); return -1; }
        cur ^= 1;
    }
    return cur;
}
