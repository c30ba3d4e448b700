// exchange.hip — multi-GPU step between stage 2 and stage 3: the pixel segments a rank rasterized
// from ITS share of the lines are bucketed by the rank that owns their tile row, so that ONE all-to-all over xGMI
// delivers them, and the owner gathers what it received into one stream for its sort and painter.
//
// `tile_y` is the most significant key field and the cover carry never crosses tile rows (reference
// forma/src/cpu/painter/mod.rs:518-522, 741-776: rows are the CPU backend's unit of parallelism too), so a rank that
// holds every segment of a band of rows, in global line order, produces exactly the rows a single device would.
// Everything here is a stable partition of byte-sized work: HBM-bound, wavefront ballots for the ranks, no MFMA.
//
//   k_owner_count   : per 2048-segment block, segments per owner                       (1 read of the stream; counting
//                     inside the rasterizer instead — two ballots per 64 segments in the common one-owner case — was
//                     measured: rasterizer +65 us for the 29 us this kernel takes, dropped)
//   k_owner_scan    : per owner, exclusive scan of the block counts; totals -> the bucket HEADERS the ranks exchange with the data
//   k_owner_scatter : stable scatter into the send buffer, bucket g at [g * (C + 1), g * (C + 1) + count_g)
// A bucket is C data words + ONE header word {count: 32 | sender overflowed: bit 32} at index C, so that ONE equal-split
// all-to-all of (C + 1) words per pair moves everything (round 2 exchanged the counts in a collective of their own: a second
// RCCL launch per frame, ~70 us with a world of one).
//   k_gather_chunks : received buckets (rank-major = global line order) -> one contiguous stream, its length, the
//                     varying-bit masks of its keys and whether it is non-decreasing in layer (what the sort plan needs)
#include "common.h"

#define XB_WAVES   4
#define XB_THREADS (64 * XB_WAVES)
#define XB_ROWS    8                         // segments per lane
#define XB_TILE    (XB_THREADS * XB_ROWS)    // 2048: wave w owns [512 w, 512 w + 512) of the block, row j = 64 consecutive ones

__device__ __forceinline__ uint32_t owner_of(uint64_t v, const OwnerBands& B) {
    const int ty = (int)(v >> 53) - 1;
    if (ty < (int)B.edge[0] || ty >= (int)B.edge[B.n]) return B.n;       // never painted (painter/mod.rs:731-734): dropped
    uint32_t g = 0;
#pragma unroll
    for (int k = 1; k < FORMA_MAX_RANKS; k++) g += (k < (int)B.n && ty >= (int)B.edge[k]) ? 1u : 0u;
    return g;
}

__global__ __launch_bounds__(XB_THREADS) void k_owner_count(const uint64_t* __restrict__ seg, DevCount nc, OwnerBands B,
                                                            uint32_t* __restrict__ block_counts /* [n + 1][nblocks] */,
                                                            uint32_t nblocks_cap, uint64_t* __restrict__ send, uint32_t capacity) {
    __shared__ uint32_t s_c[XB_WAVES][FORMA_MAX_RANKS + 1];
    const uint32_t n = dev_count(nc);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // the bucket headers (one word each, capacity + 1 words apart) start from zero: k_owner_scan, the next launch, ORs into them
    if (blockIdx.x == 0 && tid < (int)B.n) send[(size_t)tid * ((size_t)capacity + 1) + capacity] = 0ull;
    const uint32_t base = blockIdx.x * XB_TILE + w * (64 * XB_ROWS);
    uint32_t cnt[FORMA_MAX_RANKS + 1];
#pragma unroll
    for (int g = 0; g <= FORMA_MAX_RANKS; g++) cnt[g] = 0;
    uint64_t v[XB_ROWS];
#pragma unroll
    for (int j = 0; j < XB_ROWS; j++) { const uint32_t i = base + j * 64 + lane; v[j] = i < n ? seg[i] : 0ull; }
#pragma unroll
    for (int j = 0; j < XB_ROWS; j++) {
        const uint32_t i = base + j * 64 + lane;
        const uint32_t o = i < n ? owner_of(v[j], B) : 0xFFu;
#pragma unroll
        for (int g = 0; g <= FORMA_MAX_RANKS; g++) if (g <= (int)B.n) cnt[g] += (uint32_t)__popcll(__ballot(o == (uint32_t)g));
    }
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g <= FORMA_MAX_RANKS; g++) s_c[w][g] = cnt[g];
    }
    __syncthreads();
    if (tid <= (int)B.n) {
        uint32_t t = 0;
#pragma unroll
        for (int q = 0; q < XB_WAVES; q++) t += s_c[q][tid];
        block_counts[(size_t)tid * nblocks_cap + blockIdx.x] = t;
    }
}

// one workgroup PER OWNER g (blockIdx.x): block_counts[g][0 .. nb) -> exclusive prefix in place, total -> the count the ranks
// exchange.  A thread owns a contiguous piece of the row and has all of its loads in flight at once (the first version walked
// the row 1024 blocks at a time, one dependent global round trip per step: 15 us for 6720 blocks x 2 owners).
#define XS_PER 8
__global__ __launch_bounds__(1024) void k_owner_scan(uint32_t* __restrict__ block_counts, DevCount nc, uint32_t nblocks_cap,
                                                     uint32_t n_owners /* n + 1 */, uint32_t capacity,
                                                     uint64_t* __restrict__ send /* headers at [g (capacity + 1) + capacity], zeroed */,
                                                     FrameInfo* __restrict__ info) {
    const size_t stride = (size_t)capacity + 1;
    unsigned long long* hdr = reinterpret_cast<unsigned long long*>(send);
    __shared__ uint32_t lds[17];
    const uint32_t nb = (dev_count(nc) + XB_TILE - 1) / XB_TILE;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x;
    uint32_t* row = block_counts + (size_t)g * nblocks_cap;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += 1024 * XS_PER) {
        const uint32_t i0 = b0 + threadIdx.x * XS_PER;
        uint32_t v[XS_PER], sum = 0;
#pragma unroll
        for (int k = 0; k < XS_PER; k++) { v[k] = i0 + k < nb ? row[i0 + k] : 0u; }
#pragma unroll
        for (int k = 0; k < XS_PER; k++) sum += v[k];
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) lds[w] = inc;
        __syncthreads();
        uint32_t wb = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) { const uint32_t t = lds[q]; if (q < w) wb += t; tot += t; }
        uint32_t run = carry + wb + inc - sum;
#pragma unroll
        for (int k = 0; k < XS_PER; k++) { if (i0 + k < nb) row[i0 + k] = run; run += v[k]; }
        carry += tot;
        __syncthreads();
    }
    // (headers are zeroed before the launch and only ever OR-ed into: the count of bucket g by workgroup g, the overflow bit of
    //  every bucket by whichever workgroup finds an excess)
    if (threadIdx.x == 0 && g == 0 && nc.ptr && *nc.ptr > nc.bound) {     // read-back-free frame: more local segments than were
        for (uint32_t t = 0; t + 1 < n_owners; t++) atomicOr(&hdr[t * stride + capacity], 1ull << 32);   // provisioned — the excess was never
        info->exchange_overflow = 1u;                                    // rasterized: every receiver fails the frame, the host re-plans
    }
    if (threadIdx.x == 0 && g + 1 < n_owners) {                          // (the last owner is the dropped bucket)
        atomicOr(&hdr[g * stride + capacity], (unsigned long long)(carry < capacity ? carry : capacity));
        if (carry > capacity) {                                          // every receiver learns that this sender overflowed
            for (uint32_t t = 0; t + 1 < n_owners; t++) atomicOr(&hdr[t * stride + capacity], 1ull << 32);
            info->exchange_overflow = 1u;
        }
    }
}

__global__ __launch_bounds__(XB_THREADS) void k_owner_scatter(const uint64_t* __restrict__ seg, DevCount nc, OwnerBands B,
                                                              const uint32_t* __restrict__ block_offs, uint32_t nblocks_cap,
                                                              uint32_t capacity, uint64_t* __restrict__ send /* [n][capacity + 1] */) {
    __shared__ uint32_t s_c[XB_WAVES][FORMA_MAX_RANKS + 1];
    const uint32_t n = dev_count(nc);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t base = blockIdx.x * XB_TILE + w * (64 * XB_ROWS);
    uint64_t v[XB_ROWS];
#pragma unroll
    for (int j = 0; j < XB_ROWS; j++) { const uint32_t i = base + j * 64 + lane; v[j] = i < n ? seg[i] : 0ull; }
    uint32_t own = 0;                                                   // 4 bits per row
    uint32_t cnt[FORMA_MAX_RANKS + 1];
#pragma unroll
    for (int g = 0; g <= FORMA_MAX_RANKS; g++) cnt[g] = 0;
#pragma unroll
    for (int j = 0; j < XB_ROWS; j++) {
        const uint32_t i = base + j * 64 + lane;
        const uint32_t o = i < n ? owner_of(v[j], B) : 0xFu;
        own |= o << (4 * j);
#pragma unroll
        for (int g = 0; g < FORMA_MAX_RANKS; g++) if (g < (int)B.n) cnt[g] += (uint32_t)__popcll(__ballot(o == (uint32_t)g));
    }
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < FORMA_MAX_RANKS; g++) s_c[w][g] = cnt[g];
    }
    __syncthreads();
    // position of the wave's first segment of owner g inside bucket g: blocks before + waves before (index order = (w, j, lane))
    uint32_t run[FORMA_MAX_RANKS];
#pragma unroll
    for (int g = 0; g < FORMA_MAX_RANKS; g++) {
        uint32_t r = 0;
        if (g < (int)B.n) {
            r = block_offs[(size_t)g * nblocks_cap + blockIdx.x];
#pragma unroll
            for (int q = 0; q < XB_WAVES; q++) if (q < w) r += s_c[q][g];
        }
        run[g] = r;
    }
    const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int j = 0; j < XB_ROWS; j++) {
        const uint32_t o = (own >> (4 * j)) & 0xFu;
#pragma unroll
        for (int g = 0; g < FORMA_MAX_RANKS; g++) {
            if (g < (int)B.n) {
                const uint64_t bg = __ballot(o == (uint32_t)g);
                if (o == (uint32_t)g) {
                    const uint32_t pos = run[g] + (uint32_t)__popcll(bg & lt);
                    if (pos < capacity) send[(size_t)g * ((size_t)capacity + 1) + pos] = v[j];
                }
                run[g] += (uint32_t)__popcll(bg);
            }
        }
    }
}

// received buckets (chunk s = what rank s sent: [s * (capacity + 1), + count_s), header at + capacity), rank-major, -> out[0 .. N_r).
// blockIdx.y = chunk: every workgroup copies a contiguous piece of ONE chunk (coalesced 8-byte loads and stores).
__global__ __launch_bounds__(256) void k_gather_chunks(const uint64_t* __restrict__ recv,
                                                       uint32_t n_ranks, uint32_t capacity, uint64_t* __restrict__ out,
                                                       FrameInfo* __restrict__ info, uint32_t* __restrict__ mask_records) {
    __shared__ uint32_t red[5][4];
    const uint32_t s = blockIdx.y;
    uint32_t start = 0, total = 0, over = 0, prev_chunk = 0xFFFFFFFFu, prev_cnt = 0;
    const size_t stride = (size_t)capacity + 1;
    uint32_t cnt_s = 0;
    for (uint32_t q = 0; q < n_ranks; q++) {                            // (uniform: <= 8 scalar loads)
        const uint64_t h = recv[q * stride + capacity];
        const uint32_t c0 = (uint32_t)h;
        if (q == s) cnt_s = c0;
        over |= (uint32_t)(h >> 32) | (c0 > capacity ? 1u : 0u);
        const uint32_t c = c0 < capacity ? c0 : capacity;
        if (q < s) { start += c; if (c) { prev_chunk = q; prev_cnt = c; } }
        total += c;
    }
    const uint32_t cnt = cnt_s < capacity ? cnt_s : capacity;
    if (blockIdx.x == 0 && s == 0 && threadIdx.x == 0) { info->n_segments = total; if (over) { info->exchange_overflow = 1u; info->plan_bad = 1u; } }
    const uint64_t* src = recv + (size_t)s * stride;
    uint32_t k_or = 0, k_or_hi = 0, k_and = 0xFFFFFFFFu, k_and_hi = 0xFFFFFFFFu, unsorted = 0;
    for (uint32_t k0 = blockIdx.x * 1024; k0 < cnt; k0 += gridDim.x * 1024) {
        uint64_t v[4], pv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t k = k0 + j * 256 + threadIdx.x;
            v[j] = k < cnt ? src[k] : 0ull;
            pv[j] = ((threadIdx.x & 63) == 0 && k < cnt && k > 0) ? src[k - 1] : 0ull;   // lane 0: the element in front of the wave's row
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint64_t up = __shfl_up(v[j], 1, 64); if (threadIdx.x & 63) pv[j] = up; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t k = k0 + j * 256 + threadIdx.x;
            if (k < cnt) {
                out[start + k] = v[j];
                const uint32_t klo = (uint32_t)(v[j] >> 20), khi = (uint32_t)(v[j] >> 52);
                k_or |= klo; k_or_hi |= khi; k_and &= klo; k_and_hi &= khi;
                uint64_t p = pv[j];
                if (k == 0) p = prev_chunk != 0xFFFFFFFFu ? recv[(size_t)prev_chunk * stride + prev_cnt - 1] : v[j];
                if (seg_layer(p) > seg_layer(v[j])) unsorted = 1;       // is the gathered stream non-decreasing in layer?
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        k_or |= __shfl_xor(k_or, d, 64); k_or_hi |= __shfl_xor(k_or_hi, d, 64);
        k_and &= __shfl_xor(k_and, d, 64); k_and_hi &= __shfl_xor(k_and_hi, d, 64);
        unsorted |= __shfl_xor(unsorted, d, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = k_or; red[1][w] = k_or_hi; red[2][w] = k_and; red[3][w] = k_and_hi; red[4][w] = unsorted; }
    __syncthreads();
    if (threadIdx.x == 0) {                                             // one record per workgroup (neutral if it copied nothing),
        uint32_t o = 0, oh = 0, a = 0xFFFFFFFFu, ah = 0xFFFFFFFFu, u = 0; // combined by k_reduce_gather_masks: see k_rasterize
        for (int q = 0; q < 4; q++) { o |= red[0][q]; oh |= red[1][q]; a &= red[2][q]; ah &= red[3][q]; u |= red[4][q]; }
        uint32_t* m = mask_records + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        m[0] = o; m[1] = oh; m[2] = a; m[3] = ah; m[4] = u;
    }
}

__global__ __launch_bounds__(1024) void k_reduce_gather_masks(const uint32_t* __restrict__ rec, uint32_t n_records,
                                                              FrameInfo* __restrict__ info) {
    __shared__ uint32_t red[5][16];
    uint32_t o = 0, oh = 0, a = 0xFFFFFFFFu, ah = 0xFFFFFFFFu, u = 0;
    for (uint32_t b = threadIdx.x; b < n_records; b += 1024) {
        const uint4 m = *reinterpret_cast<const uint4*>(rec + (size_t)b * 8);
        o |= m.x; oh |= m.y; a &= m.z; ah &= m.w; u |= rec[(size_t)b * 8 + 4];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        o |= __shfl_xor(o, d, 64); oh |= __shfl_xor(oh, d, 64); a &= __shfl_xor(a, d, 64); ah &= __shfl_xor(ah, d, 64);
        u |= __shfl_xor(u, d, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = o; red[1][w] = oh; red[2][w] = a; red[3][w] = ah; red[4][w] = u; }
    __syncthreads();
    if (threadIdx.x == 0 && info->n_segments) {                         // (an empty band keeps the reset masks)
        for (int i = 1; i < 16; i++) { o |= red[0][i]; oh |= red[1][i]; a &= red[2][i]; ah &= red[3][i]; u |= red[4][i]; }
        info->key_or = o; info->key_or_hi = oh; info->key_and = a; info->key_and_hi = ah; info->layer_unsorted = u;
    }
}

// pixel segments per tile row (planning of a multi-device context: band edges and pair capacities come from it).  Row r
// of the canvas is bin r; bin 2047 collects what is never painted (tile_y < 0 stored as row 0 - 1, rows >= 2047).  A lane
// walks 8 consecutive segments and adds whole runs (consecutive segments mostly share their row).
__global__ __launch_bounds__(256) void k_row_histogram(const uint64_t* __restrict__ seg, DevCount nc, uint32_t* __restrict__ hist) {
    __shared__ uint32_t lh[2048];
    const uint32_t n = dev_count(nc);
    for (int i = threadIdx.x; i < 2048; i += 256) lh[i] = 0;
    __syncthreads();
    for (uint32_t base = (blockIdx.x * 256 + threadIdx.x) * 8; base < n; base += gridDim.x * 256 * 8) {
        uint32_t cur = 0xFFFFFFFFu, run = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (base + q >= n) break;
            const uint32_t tyb = (uint32_t)(seg[base + q] >> 53);
            const uint32_t bin = tyb >= 1u && tyb <= 2047u ? tyb - 1u : 2047u;
            if (bin != cur) { if (run) atomicAdd(&lh[cur], run); cur = bin; run = 0; }
            run++;
        }
        if (run) atomicAdd(&lh[cur], run);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256) if (lh[i]) atomicAdd(&hist[i], lh[i]);
}
void launch_row_histogram(hipStream_t s, const uint64_t* seg, DevCount n, uint32_t* hist /* 2048 words, zeroed here */) {
    (void)hipMemsetAsync(hist, 0, 2048 * 4, s);
    if (n.bound == 0) return;
    uint32_t blocks = (n.bound + 2047) / 2048;
    if (blocks > 1024) blocks = 1024;
    FORMA_LAUNCH(k_row_histogram, dim3(blocks), dim3(256), 0, s, seg, n, hist);
}

size_t owner_scratch_words(size_t n) { return (size_t)(FORMA_MAX_RANKS + 1) * ((n + XB_TILE - 1) / XB_TILE + 1); }

void launch_owner_bucket(hipStream_t s, const uint64_t* seg, DevCount nc, const OwnerBands& B, uint32_t capacity,
                         uint32_t* scratch, uint64_t* send, FrameInfo* info) {
    const uint32_t nblocks = (uint32_t)((nc.bound + XB_TILE - 1) / XB_TILE);
    const uint32_t cap = nblocks + 1;
    if (nblocks == 0) {                                   // (no kernel: the headers are cleared here)
        (void)hipMemset2DAsync(send + capacity, ((size_t)capacity + 1) * 8, 0, 8, B.n, s);
        return;
    }
    FORMA_LAUNCH(k_owner_count, dim3(nblocks), dim3(XB_THREADS), 0, s, seg, nc, B, scratch, cap, send, capacity);
    FORMA_LAUNCH(k_owner_scan, dim3(B.n + 1), dim3(1024), 0, s, scratch, nc, cap, B.n + 1, capacity, send, info);
    FORMA_LAUNCH(k_owner_scatter, dim3(nblocks), dim3(XB_THREADS), 0, s, seg, nc, B, (const uint32_t*)scratch, cap, capacity, send);
}

size_t gather_mask_words(uint32_t n_ranks, uint32_t capacity) {
    uint32_t gx = (capacity + 1023) / 1024;
    if (gx > 2048) gx = 2048;
    if (gx == 0) gx = 1;
    return (size_t)gx * n_ranks * 8;
}

void launch_gather_chunks(hipStream_t s, const uint64_t* recv, uint32_t n_ranks, uint32_t capacity,
                          uint64_t* out, FrameInfo* info, uint32_t* mask_records, bool reduce_now) {
    uint32_t gx = (capacity + 1023) / 1024;
    if (gx > 2048) gx = 2048;
    if (gx == 0) gx = 1;
    FORMA_LAUNCH(k_gather_chunks, dim3(gx, n_ranks), dim3(256), 0, s, recv, n_ranks, capacity, out, info, mask_records);
    if (reduce_now) FORMA_LAUNCH(k_reduce_gather_masks, dim3(1), dim3(1024), 0, s, (const uint32_t*)mask_records, gx * n_ranks, info);
}
