// mgpu_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the shard-query path.
//
// All kernels are HBM/L2-bound integer work: 8-16 B per lane coalesced streams over dense
// u64 bitmap rows, wave-level popcount / ballot compaction, LDS-privatised histograms and an
// LDS bitonic stage for the final top-k.  No MFMA: nothing here is a dense contraction.
// blockIdx.y always selects the (shard, query) pair ("squery") of a batch.
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include "mgpu_types.h"
#include "mg_lev.h"

typedef unsigned long long u64;
typedef unsigned __int128 u128;

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ u64 wave_or64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ bool key_lt(u64 ah, u64 al, uint32_t ai, u64 bh, u64 bl, uint32_t bi) {
    // (no short circuits: the operands are loaded and compared unconditionally, so loops over LDS pipeline their loads)
    return (ah < bh) | ((ah == bh) & ((al < bl) | ((al == bl) & (ai < bi))));
}
// ---- optional phase clocks (mgpu_debug_phases): cycles summed per phase over the active workgroups
__device__ int mg_dbg_on = 0;
__device__ int mg_dbg_skip = 0;        // diagnostic: bit 2 = k_rank_tile without phase 2, bit 4 = without probing (tools/rank_parts.py); keys are then wrong, nothing else is
__device__ unsigned long long mg_dbg_acc[32];
#define DBG_T(var) const long long var = mg_dbg_on ? (long long)wall_clock64() : 0
#define DBG_ADD(slot, a, b) do { if (mg_dbg_on && threadIdx.x == 0) atomicAdd(&mg_dbg_acc[slot], (unsigned long long)((b) - (a))); } while (0)
#define DBG_MAX(slot, a, b, tag) do { if (mg_dbg_on && threadIdx.x == 0) atomicMax(&mg_dbg_acc[slot], ((unsigned long long)((b) - (a)) << 32) | (unsigned long long)(tag)); } while (0)
struct DbgScope {       // records the longest lifetime of a workgroup (slot) with a tag, at every exit path
    long long t0; int slot; unsigned tag;
    __device__ __forceinline__ DbgScope(int s) : t0(mg_dbg_on ? (long long)wall_clock64() : 0), slot(s), tag(0) {}
    __device__ __forceinline__ ~DbgScope() { if (mg_dbg_on && threadIdx.x == 0) { long long t1 = (long long)wall_clock64(); atomicMax(&mg_dbg_acc[slot], ((unsigned long long)(t1 - t0) << 32) | tag);
        atomicMax(&mg_dbg_acc[14], ~(unsigned long long)t0); atomicMax(&mg_dbg_acc[15], (unsigned long long)t1); } }
};
#define DBG_CNT(slot) do { if (mg_dbg_on && threadIdx.x == 0) atomicAdd(&mg_dbg_acc[slot], 1ull); } while (0)

// exclusive scan of one u32 per thread over a 256-thread block; returns the block total in *total
template <int NT = MG_BLOCK>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *total, uint32_t *lds /* >= NT / 64 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
        uint32_t s = lds[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// ------------------------------------------------------------------ scatter (sparse postings -> rows)
// oper_or's array half (cont.c:431-435 array_cont_inplace_union) and num_*_filter's bmap_add
// loop (filter_apply.c:44-54,159-166,182-191): sorted docid lists OR-ed into a zeroed dense row.
// Lanes that hit the same u64 word are merged in-wave first, so one atomic goes out per word.
__device__ __forceinline__ void scatter_chunk(const uint32_t *src, uint32_t n, u64 *dst) {
    for (uint32_t base = 0; base < n; base += MG_BLOCK) {
        uint32_t i = base + threadIdx.x;
        bool valid = i < n;
        uint32_t d = valid ? src[i] : 0xFFFFFFFFu;
        uint32_t w = d >> 6;
        u64 bit = valid ? (1ull << (d & 63)) : 0ull;
        // sorted input: equal words are adjacent lanes; segmented OR by doubling
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t ow = __shfl_down(w, o, 64);
            u64 ob = __shfl_down(bit, o, 64);
            if (lane + o < 64 && ow == w) bit |= ob;
        }
        uint32_t pw = __shfl_up(w, 1, 64);
        bool head = (lane == 0) || (pw != w);
        if (valid && head) atomicOr(&dst[w], bit);
    }
}
__global__ __launch_bounds__(MG_BLOCK) void k_scatter(const ScatterItem *items, uint32_t nitems) {
    for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const ScatterItem s = items[it];
        scatter_chunk(s.src, s.n, (u64 *)s.dst);
    }
}
// RangeRow (mgpu_types.h): grid (x, groups of MG_RANGE_GROUP rows); a wave takes 64 words = 4096 documents at a time, 64
// coalesced loads of the column per lane, one ballot each; lane i keeps word i.  Every word of the row below W is written (no
// zeroing needed).  Rows of one group that read the SAME column (the host lists the rows by column: a batch of queries
// with a range each on one field) share its loads: the column is read once per group, not once per row -- the rows
// of a batch re-read it from L2 / the memory-side cache otherwise, and that traffic was the kernel (0.90 ms per 1024 rows;
// 0.38 ms with groups of eight, where the two compares and the two selects per (word, row) are what is left: groups of 16
// read 0.40 ms).
#define MG_RANGE_GROUP 8
__device__ __forceinline__ bool range_row_bounds(const RangeRow &r, double *v_lo, double *v_hi) {
    if (r.lo >= r.hi) return false;
    const uint32_t da = r.order[r.lo], db = r.order[r.hi - 1];
    if (da >= r.N || db >= r.N) return false;                         // (the mirror's own tables: cannot happen)
    const double va = r.col[da], vb = r.col[db];
    *v_lo = va < vb ? va : vb; *v_hi = va < vb ? vb : va;
    return true;
}
__global__ __launch_bounds__(MG_BLOCK) void k_range_rows(const RangeRow *rows, uint32_t nrows) {
    const uint32_t first = blockIdx.y * MG_RANGE_GROUP;
    const uint32_t cnt = nrows - first < MG_RANGE_GROUP ? nrows - first : MG_RANGE_GROUP;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * MG_BLOCK + threadIdx.x) >> 6, nwaves = gridDim.x * (MG_BLOCK / 64);
    RangeRow r[MG_RANGE_GROUP];
    double v_lo[MG_RANGE_GROUP], v_hi[MG_RANGE_GROUP];
    bool ok[MG_RANGE_GROUP], same = true;
#pragma unroll
    for (int g = 0; g < MG_RANGE_GROUP; g++) {
        r[g] = rows[first + ((uint32_t)g < cnt ? g : 0)];
        ok[g] = (uint32_t)g < cnt && range_row_bounds(r[g], &v_lo[g], &v_hi[g]);
        if (!ok[g]) { v_lo[g] = 1.0; v_hi[g] = 0.0; }                  // matches nothing
        same = same && r[g].col == r[0].col && r[g].N == r[0].N && r[g].W == r[0].W;
    }
    if (same) {
        const uint32_t N = r[0].N, W = r[0].W;
        const double *col = r[0].col;
        for (uint32_t wb = wave * 64; wb < W; wb += nwaves * 64) {    // wave-uniform
            u64 mine[MG_RANGE_GROUP];
#pragma unroll
            for (int g = 0; g < MG_RANGE_GROUP; g++) mine[g] = 0;
#pragma unroll 4
            for (uint32_t i = 0; i < 64; i++) {
                const uint32_t d = (wb + i) * 64 + lane;
                const double v = d < N ? col[d] : -1.7976931348623157e308;           // "no value" (get_num_value's marker)
#pragma unroll
                for (int g = 0; g < MG_RANGE_GROUP; g++) {
                    const u64 m = __ballot(v >= v_lo[g] && v <= v_hi[g]);
                    if (lane == i) mine[g] = m;
                }
            }
#pragma unroll
            for (int g = 0; g < MG_RANGE_GROUP; g++) if (ok[g] && wb + lane < W) r[g].dst[wb + lane] = mine[g];
        }
        return;
    }
    for (uint32_t g = 0; g < cnt; g++) {                              // rows of different columns: one at a time
        double lo, hi;
        const RangeRow q = rows[first + g];
        if (!range_row_bounds(q, &lo, &hi)) continue;
        for (uint32_t wb = wave * 64; wb < q.W; wb += nwaves * 64) {
            u64 mine = 0;
#pragma unroll 8
            for (uint32_t i = 0; i < 64; i++) {
                const uint32_t d = (wb + i) * 64 + lane;
                const double v = d < q.N ? q.col[d] : -1.7976931348623157e308;
                const u64 m = __ballot(v >= lo && v <= hi);
                if (lane == i) mine = m;
            }
            if (wb + lane < q.W) q.dst[wb + lane] = mine;
        }
    }
}
// the same for ONE list named by the kernel arguments (no item table to upload): chunks of MG_SCATTER_CHUNK docids
__global__ __launch_bounds__(MG_BLOCK) void k_scatter_list(const uint32_t *src, uint64_t n, u64 *dst) {
    const uint64_t nchunks = (n + MG_SCATTER_CHUNK - 1) / MG_SCATTER_CHUNK;
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint64_t at = c * MG_SCATTER_CHUNK;
        const uint64_t left = n - at;
        scatter_chunk(src + at, (uint32_t)(left < MG_SCATTER_CHUNK ? left : MG_SCATTER_CHUNK), dst);
    }
}

// ------------------------------------------------------------------ formula + filter + popcount
// get_matching_docids (squery.c:212-277) fused with squery_apply_filters (squery.c:377-395,
// filter_apply.c:207-227) and bmap_cardinality (bmap.c:362-368).
// A workgroup owns one window = MG_FORMULA_WORDS consecutive u64 words (32768 docids) of the squery's
// rows; lane t owns words 2t, 2t+1 of it.  Every term gets a tile of the window in LDS:
//   dense postings: the lane ORs its 16 bytes of each dense row of the term into its words of the
//     tile (one flat list of {row, term} entries, loads batched four deep, no synchronisation);
//   sparse postings never become rows in memory: the list table gives {term, first element, length,
//     window index}, one lane per list looks up its [start, end) for this window (post_win; short
//     lists are taken whole and filtered by docid), a block scan turns the lengths into one flat
//     element range and the lanes stride over it with LDS atomic ORs -- every element load of the
//     workgroup is in flight at once.
// The match formula and the filter program then run on the tiles in registers.
#define MG_PLAN_LDS 1024
#define MG_TILE_TERMS 6              // term tiles held in LDS at once (24 KiB): every term of a query of <= 3 words
struct W2 { u64 a, b; };
__device__ __forceinline__ W2 w2_or(W2 x, W2 y) { return {x.a | y.a, x.b | y.b}; }
__device__ __forceinline__ W2 w2_and(W2 x, W2 y) { return {x.a & y.a, x.b & y.b}; }
__device__ __forceinline__ W2 w2_andnot(W2 x, W2 y) { return {x.a & ~y.a, x.b & ~y.b}; }
__device__ __forceinline__ W2 w2_load(const u64 *row, uint32_t w0, bool in0) {
    // rows are padded to a multiple of 32 words (Wp), so a pair of words never straddles the end
    const ulonglong2 x = in0 ? *(const ulonglong2 *)(row + w0) : make_ulonglong2(0ull, 0ull);
    return {x.x, x.y};
}

__device__ __forceinline__ void emit_bits(u64 v, uint32_t base, uint32_t *hits, uint32_t &off) {
    while (v) {
        const int b = __ffsll((long long)v) - 1;
        v &= v - 1;
        hits[off++] = base + (uint32_t)b;
    }
}
#define MG_FORMULA_RUN 4             // consecutive windows of a squery that one workgroup takes when the launch is large
// (five waves per SIMD is what the 31 KB of LDS allow; unbounded, the compiler takes 103 VGPRs for the loop and leaves four)
__global__ __launch_bounds__(MG_BLOCK) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_formula(const SqDev *sqs, const uint32_t *blob, uint32_t run) {
    const SqDev *sp = &sqs[blockIdx.y];
    // the zero-typo twin of a squery that is not going to be ranked on a sample (k_order of the squeries themselves has run
    // and set the flag: mgk_formula): nobody reads its map or its count
    if (sp->aux && sp->fast_flag && !*sp->fast_flag) return;
    const uint32_t nblk = sp->nblk;                                  // 256-word blocks of the row
    const uint32_t nwin = (nblk + MG_FORMULA_WORDS / MG_WORDS_PER_BLOCK - 1) / (MG_FORMULA_WORDS / MG_WORDS_PER_BLOCK);
    // A workgroup takes a run of consecutive windows: the descriptor, the plan and the descriptors of the first 256
    // lists are fetched once per run, and the next window's slice bounds travel behind the current window's emit.
    const uint32_t win_first = blockIdx.x * run;
    if (win_first >= nwin) return;
    const uint32_t win_end = win_first + run < nwin ? win_first + run : nwin;
    // descriptor fields in one burst
    const uint32_t Wp = sp->sh.Wp;
    const int nt = sp->num_terms, nf = sp->num_fops, empty = sp->empty;
    const uint32_t plan_off = sp->plan_off, plan_len = sp->plan_len, plan_fops = sp->plan_fops;
    const u64 *dense = (const u64 *)sp->sh.dense;
    const u64 *rows = (const u64 *)sp->rows;
    const uint32_t *post_doc = sp->sh.post_doc, *post_win = sp->sh.post_win;
    const uint4 *sl = (const uint4 *)(blob + sp->sl_off);
    u64 *docid_map = (u64 *)sp->docid_map;
    uint2 *win_info = sp->win_info;
    uint32_t *hits_raw = sp->hits_raw, *raw_count = sp->raw_count, *err = sp->err;
    const uint32_t hit_cap = sp->hit_cap;
    __shared__ uint32_t s_plan[MG_PLAN_LDS];
    __shared__ uint32_t red[4];
    __shared__ __align__(16) u64 s_tile[MG_TILE_TERMS * MG_FORMULA_WORDS];
    __shared__ uint32_t s_pre[MG_BLOCK], s_src[MG_BLOCK], s_trm[MG_BLOCK], s_scan[8];
    const uint32_t *plan = blob + plan_off;
    const bool in_lds = plan_len <= MG_PLAN_LDS;
    // first term group's lists: descriptor and window range of list (lb0 + lane) go out now, together
    // with the plan, so that only the element loads remain behind the first barrier
    const int tg0 = nt < MG_TILE_TERMS ? nt : MG_TILE_TERMS;
    const uint32_t lb0 = sp->term_sl[0], le0 = sp->term_sl[tg0];
    uint4 d0 = make_uint4(0, 0, 0, 0xFFFFFFFFu);
    uint32_t r0s = 0, r0e = 0;
    if (!empty && lb0 + threadIdx.x < le0) {
        d0 = sl[lb0 + threadIdx.x];
        r0e = d0.z;
        if (d0.w != 0xFFFFFFFFu) { r0s = post_win[d0.w + win_first]; r0e = post_win[d0.w + win_first + 1]; }
    }
    if (in_lds) {
        for (uint32_t i = threadIdx.x; i < plan_len; i += MG_BLOCK) s_plan[i] = plan[i];
        __syncthreads();
    }
    auto word = [&](uint32_t i) -> uint32_t { return in_lds ? s_plan[i] : plan[i]; };
    for (uint32_t win = win_first; win < win_end; win++) {
    const uint32_t base = win * MG_FORMULA_WORDS;
    const uint32_t w0 = base + 2 * threadIdx.x;
    const bool in0 = w0 < Wp;
    W2 v = {0, 0};
    if (!empty) {                                 // uniform: the tile builder below synchronises the workgroup
        int tcur = 0;
        auto row_of = [&](uint32_t e) -> const u64 * {
            return ((int32_t)e >= 0) ? dense + (size_t)e * Wp : rows + (size_t)(~e) * Wp;
        };
        auto build_tiles = [&](int t0) {
            const int t1 = (t0 + MG_TILE_TERMS < nt) ? t0 + MG_TILE_TERMS : nt;
            if (t0) __syncthreads();              // the previous group's tiles have been read
            // ---- dense rows: this lane's words of each tile, nobody else touches them yet
            u64 *mine = s_tile + 2 * threadIdx.x;
            for (int j = 0; j < t1 - t0; j++) *(ulonglong2 *)(mine + j * MG_FORMULA_WORDS) = make_ulonglong2(0ull, 0ull);
            const uint32_t dn0 = word((uint32_t)t0), dn1 = word((uint32_t)t1), lbase = (uint32_t)nt + 1;
            for (uint32_t e = dn0; e < dn1; e += 4) {
                uint32_t en[4]; ulonglong2 x[4];
#pragma unroll
                for (int u = 0; u < 4; u++) en[u] = (e + u < dn1) ? word(lbase + e + u) : 0xFFFFFFFFu;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    x[u] = (en[u] != 0xFFFFFFFFu && in0) ? *(const ulonglong2 *)(dense + (size_t)(en[u] & 0x7FFFFFFu) * Wp + w0)
                                                         : make_ulonglong2(0ull, 0ull);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (en[u] != 0xFFFFFFFFu) {
                        u64 *tl = mine + ((en[u] >> 27) - (uint32_t)t0) * MG_FORMULA_WORDS;
                        tl[0] |= x[u].x; tl[1] |= x[u].y;
                    }
                }
            }
            // ---- sparse lists
            const uint32_t lb = t0 ? sp->term_sl[t0] : lb0, le = t0 ? sp->term_sl[t1] : le0;
            for (uint32_t c = lb; c < le; c += MG_BLOCK) {
                const uint32_t k = c + threadIdx.x;
                uint32_t cnt = 0, src = 0, trm = 0;
                if (k < le) {
                    uint4 d = d0;                 // {term, first element, length, window index base}
                    uint32_t s0 = r0s, e0 = r0e;
                    if (t0 || c != lb) {
                        d = sl[k];
                        s0 = 0; e0 = d.z;
                        if (d.w != 0xFFFFFFFFu) { s0 = post_win[d.w + win]; e0 = post_win[d.w + win + 1]; }
                    }
                    cnt = e0 - s0; src = d.y + s0;
                    trm = (d.x - (uint32_t)t0) | (d.w == 0xFFFFFFFFu ? 0x100u : 0u);
                }
                uint32_t tot;
                const uint32_t excl = block_excl_scan(cnt, &tot, s_scan);
                s_pre[threadIdx.x] = (k < le) ? excl : 0xFFFFFFFFu;
                s_src[threadIdx.x] = src; s_trm[threadIdx.x] = trm;
                __syncthreads();                  // (also: every lane's dense words are in the tiles before the first OR)
                for (uint32_t x = threadIdx.x; x < tot; x += MG_BLOCK) {
                    // the list holding flat element x: the last one whose first element is <= x
                    uint32_t j = 0;
#pragma unroll
                    for (uint32_t st = MG_BLOCK / 2; st > 0; st >>= 1) if (s_pre[j + st] <= x) j += st;
                    const uint32_t d = post_doc[s_src[j] + (x - s_pre[j])];
                    const uint32_t tr = s_trm[j];
                    const uint32_t w = d >> 6;
                    if ((tr & 0x100u) && (w / MG_FORMULA_WORDS) != win) continue;      // short list: other windows' docids
                    // (32-bit halves of the little-endian u64 words)
                    atomicOr((uint32_t *)s_tile + ((tr & 0xFFu) * MG_FORMULA_WORDS + (w - base)) * 2 + ((d >> 5) & 1u), 1u << (d & 31));
                }
                __syncthreads();
            }
        };
        auto term = [&]() -> W2 {
            if (tcur % MG_TILE_TERMS == 0) build_tiles(tcur);
            const ulonglong2 x = *(const ulonglong2 *)(s_tile + (tcur % MG_TILE_TERMS) * MG_FORMULA_WORDS + 2 * threadIdx.x);
            tcur++;
            return {x.x, x.y};
        };
        if (nt == 1) v = term();
        else if (nt == 3) { const W2 t0 = term(), t1 = term(), t2 = term(); v = w2_or(w2_and(t0, t1), t2); }
        else {
            const W2 ones = {~0ull, ~0ull};
            W2 acc = ones, prev = {0, 0};
            for (int i = 0; i < nt; i += 2) {
                W2 g = w2_or(prev, term());
                W2 next = {0, 0};
                if (i < nt - 2) { next = term(); g = w2_or(g, next); }
                acc = w2_and(acc, g);
                prev = next;
            }
            v = w2_or(acc, term());
        }
        if (nf > 0) {
            // postfix filter program on a small register stack (static indexing only)
            const W2 z = {0, 0};
            W2 s0 = z, s1 = z, s2 = z, s3 = z, s4 = z, s5 = z, s6 = z, s7 = z;
            for (int i = 0; i < nf; i++) {
                const int op = (int)word(plan_fops + 2 * i);
                const uint32_t a = word(plan_fops + 2 * i + 1);
                if (op <= MGPU_F_PUSH_EMPTY) {
                    W2 x = z;
                    if (op != MGPU_F_PUSH_EMPTY) x = w2_load(row_of(a), w0, in0);
                    s7 = s6; s6 = s5; s5 = s4; s4 = s3; s3 = s2; s2 = s1; s1 = s0; s0 = x;
                } else if (op == MGPU_F_DOCS_ANDNOT) {
                    s0 = w2_andnot(v, s0);
                } else {
                    s0 = (op == MGPU_F_AND) ? w2_and(s1, s0) : w2_or(s1, s0);
                    s1 = s2; s2 = s3; s3 = s4; s4 = s5; s5 = s6; s6 = s7; s7 = z;
                }
            }
            v = w2_and(v, s0);
        }
        if (!in0) { v.a = 0; v.b = 0; }          // words past the row (the last window of a shard) hold nothing
    }
    // the next window's slice of this lane's list (its start is this window's end): in flight behind the emit
    uint32_t r0e_next = r0e;
    if (win + 1 < win_end && d0.w != 0xFFFFFFFFu) r0e_next = post_win[d0.w + win + 2];
    // (the dense docid_map is only written when somebody asked for it: parity captures)
    if (docid_map && in0) *(ulonglong2 *)(docid_map + w0) = make_ulonglong2(v.a, v.b);
    // bmap_iterate fused in (bmap.c:370-374, cont.c:590-610): the window's hits go straight to the raw hit list.
    // The workgroup reserves its range with ONE atomic on the squery's counter, so the windows of a squery land in
    // whatever order they finish; k_order puts them in window (= docid) order afterwards from {count, base}.
    const uint32_t cnt = (uint32_t)(__popcll(v.a) + __popcll(v.b));
    uint32_t tot;
    uint32_t off = block_excl_scan(cnt, &tot, s_scan);
    if (threadIdx.x == 0) {
        uint32_t b = 0;
        if (tot && hits_raw) {                      // (a zero-typo twin has no hit list: its bitmap and count are all that is wanted)
            b = atomicAdd(raw_count, tot);
            if (b + tot > hit_cap) { atomicMax(err, (uint32_t)(-MGPU_E_LIMIT)); b = 0xFFFFFFFFu; }   // the host's bound was wrong
        }
        red[0] = b;
        win_info[win] = make_uint2(tot, b);
    }
    __syncthreads();
    const uint32_t rbase = red[0];
    if (cnt && hits_raw && rbase != 0xFFFFFFFFu) {
        off += rbase;
        emit_bits(v.a, w0 << 6, hits_raw, off); emit_bits(v.b, (w0 + 1) << 6, hits_raw, off);
    }
    if (d0.w != 0xFFFFFFFFu) { r0s = r0e; r0e = r0e_next; }
    }
}

// One workgroup per squery, after k_formula: exclusive scan of the windows' hit counts in window order, the squery's
// hit count, and the raw hit list copied into ascending docid order (a window's hits are ascending and contiguous in
// the raw list, at the base its workgroup was given).  Also publishes the squery's hit tiles (256 hits each) on the
// batch-wide work list that the rank and facet/agg kernels stride over: their grids never depend on a host-side guess
// of the hit count, and a batch of uneven queries balances itself.
// fh_list / fh_count: the select jobs k_facet_hash takes, by size class ([0] up to MG_FHASH_SMALL_M hits: first fh_cap
// entries, [1] up to MG_FHASH_MAX_M: the next fh_cap); a squery names its candidates in SqDev::fhash_jobs
__global__ __launch_bounds__(MG_BLOCK) void k_order(const SqDev *sqs, uint2 *tile_list, uint32_t *tile_count,
                                                    uint32_t *fh_list, uint32_t *fh_count, uint32_t fh_cap) {
    const SqDev *sp = &sqs[blockIdx.x];
    const uint32_t nwin = (sp->nblk + 1) / 2;
    uint2 *win_info = sp->win_info;
    const uint32_t hit_cap = sp->hit_cap;
    uint32_t *p_err = sp->err, *p_num_hits = sp->num_hits;
    if (sp->aux && sp->fast_flag && !*sp->fast_flag) {               // a twin k_formula skipped: no windows to add up
        if (threadIdx.x == 0) *p_num_hits = 0;
        return;
    }
    __shared__ uint32_t lds[8];
    __shared__ uint32_t carry, s_tbase;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nwin; base += MG_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint2 wi = (i < nwin) ? win_info[i] : make_uint2(0u, 0u);
        uint32_t tot;
        const uint32_t ex = block_excl_scan(wi.x, &tot, lds);
        const uint32_t c = carry;
        if (i < nwin) win_info[i] = make_uint2(c + ex, wi.y);       // {ordered base, raw base}: what k_order_copy reads
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    uint32_t M = carry;
    if (sp->aux) {                       // zero-typo twin: the count is the result
        if (threadIdx.x == 0) *p_num_hits = M;
        return;
    }
    if (M > hit_cap || *p_err) {         // the host's bound was wrong (k_formula refused to write): report, touch no per-hit array
        if (threadIdx.x == 0) atomicMax(p_err, (uint32_t)(-MGPU_E_LIMIT));
        M = 0;
    }
    const uint32_t ntiles = (M + MG_BLOCK - 1) / MG_BLOCK;
    if (threadIdx.x == 0) {
        *p_num_hits = M; s_tbase = ntiles ? atomicAdd(tile_count, ntiles) : 0;
        if (sp->fast_eligible) {         // perform_ranking's switch (docrank.c:571)
            *sp->total_hits = M; *sp->num_rank = M;
            *sp->fast_flag = M > sp->fast_thresh ? 1u : 0u;
        }
    }
    // (a squery that may be ranked on a sample names hash jobs only when its threshold is at least MG_FHASH_MAX_M -- the host's
    // rule: "few enough hits for hashing" then implies "not sampled", for this kernel, the rank kernel and k_select_first alike)
    const uint32_t nfh = sp->fhash_njobs;
    if (threadIdx.x < nfh && M <= MG_FHASH_MAX_M) {
        const uint32_t cls = M <= MG_FHASH_SMALL_M ? 0u : 1u;
        const uint32_t at = atomicAdd(&fh_count[cls], 1u);
        if (at < fh_cap) fh_list[cls * fh_cap + at] = sp->fhash_jobs[threadIdx.x];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ntiles; i += MG_BLOCK) tile_list[s_tbase + i] = make_uint2(blockIdx.x, i);
}

// the copy itself, one tile of 256 ordered positions per work item (a squery with many hits spreads over the chip)
__global__ __launch_bounds__(MG_BLOCK) void k_order_copy(const SqDev *sqs, const uint2 *tile_list, const uint32_t *tile_count) {
    __shared__ uint32_t s_obase[MG_ORDER_WINDOWS + 1], s_rbase[MG_ORDER_WINDOWS];
    const uint32_t ntiles = *tile_count;
    for (uint32_t w = blockIdx.x; w < ntiles; w += gridDim.x) {
        const uint2 item_v = tile_list[w];
        const uint2 item = make_uint2(__builtin_amdgcn_readfirstlane(item_v.x), __builtin_amdgcn_readfirstlane(item_v.y));
        const SqDev *sp = &sqs[item.x];
        const uint32_t nwin = (sp->nblk + 1) / 2;
        const uint2 *win_info = sp->win_info;
        const uint32_t *hits_raw = sp->hits_raw;
        uint32_t *hits = sp->hits;
        const uint32_t M = *sp->num_hits;
        for (uint32_t i = threadIdx.x; i < nwin; i += MG_BLOCK) { const uint2 wi = win_info[i]; s_obase[i] = wi.x; s_rbase[i] = wi.y; }
        if (threadIdx.x == 0) s_obase[nwin] = M;
        __syncthreads();
        const uint32_t x = item.y * MG_BLOCK + threadIdx.x;
        if (x < M) {
            // the window holding ordered position x: the LAST one whose ordered base is <= x (an empty window shares
            // its base with its successor)
            uint32_t lo = 0, hi = nwin;                          // invariant: s_obase[lo] <= x < s_obase[hi]
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_obase[mid] <= x) lo = mid; else hi = mid; }
            hits[x] = hits_raw[s_rbase[lo] + (x - s_obase[lo])];
        }
        __syncthreads();                                         // LDS is reused by the next work item
    }
}

// ------------------------------------------------------------------ fast-rank: the sample (docrank.c:478-560)
// perform_fast_ranking for one squery per workgroup (1024 lanes), run only when k_order set its fast_flag.  The hit
// list (all matching documents, ascending) is REPLACED by the sample S in the order the reference ranks it:
//   1. every zero-typo document when there are at most 1000 of them (ascending docids), else the zero-typo map becomes
//      the "preferred map" rmap in place of docid_map -- unfiltered, as in the reference;
//   2. documents of rmap in num_dbi[rank_by] order (reversed for ascending) not yet taken, until 1000 are in S (at
//      least one is always taken: the reference tests the count after adding);
//   3. every skip-th document of rmap in docid order that is not in S yet.
// num_hits then counts S (what the rank and facet kernels walk), num_rank = |S| - 1 (docrank.c:550: the reference drops
// the last ranked document from the sort), total_hits keeps the real hit count.
#define FB_NT 1024
__device__ __forceinline__ bool bm_bit(const u64 *m, uint32_t d) { return (m[d >> 6] >> (d & 63)) & 1ull; }
__global__ __launch_bounds__(FB_NT) void k_fast_build(const SqDev *sqs, const uint32_t *list) {
    const SqDev *sp = &sqs[list[blockIdx.x]];
    if (!*sp->fast_flag) return;
    __shared__ uint32_t lds[FB_NT / 64];
    __shared__ uint32_t s_pos, s_taken, s_run;
    const uint32_t W = sp->sh.W, N = sp->sh.N;
    uint32_t *hits = sp->hits;
    u64 *dbmap = (u64 *)sp->dbmap;
    const u64 *D = (const u64 *)sp->docid_map, *Z = (const u64 *)sp->zmap;
    if (sp->fast_bad) {
        if (threadIdx.x == 0) {
            atomicMax(sp->err, (uint32_t)(-((sp->fast_bad & 2) ? MGPU_E_LIMIT : MGPU_E_UNSUPPORTED)));
            *sp->num_hits = 0; *sp->num_rank = 0;
        }
        return;
    }
    const uint32_t total = *sp->total_hits;
    uint32_t fcount = total / 100 < 10000u ? total / 100 : 10000u;          // MIN(total / 100, MAX_HITS_LIMIT * 10)
    if (fcount < 5000u) fcount = 5000u;                                      // MAX(.., MAX_HITS_LIMIT * 5)
    const uint32_t cap = 1000u + fcount + 10u;                               // the reference's rank buffer (docrank.c:491)
    const uint32_t skip = total / fcount + 1;
    const uint32_t zt = *sp->z_count;
    const u64 *rmap = D;
    for (uint32_t w = threadIdx.x; w < W; w += FB_NT) dbmap[w] = 0;
    if (threadIdx.x == 0) { s_pos = 0; s_run = 0; }
    __syncthreads();
    // 1. zero-typo documents
    if (zt && zt <= 1000u) {
        for (uint32_t base = 0; base < W; base += FB_NT) {
            const uint32_t w = base + threadIdx.x;
            const u64 x = w < W ? Z[w] : 0ull;
            uint32_t tot;
            uint32_t off = s_pos + block_excl_scan<FB_NT>((uint32_t)__popcll(x), &tot, lds);
            if (x) { dbmap[w] = x; emit_bits(x, w << 6, hits, off); }
            __syncthreads();
            if (threadIdx.x == 0) s_pos += tot;
            __syncthreads();
        }
    } else if (zt) rmap = Z;
    __threadfence();
    __syncthreads();
    // 2. ranking order
    {
        const uint32_t *np = sp->np_rank; const uint64_t n = sp->np_rank_n;
        const uint32_t r0 = s_pos;
        const uint32_t take = r0 < 1000u ? 1000u - r0 : 1u;
        if (threadIdx.x == 0) s_taken = 0;
        __syncthreads();
        for (uint64_t base = 0; base < n; base += 4 * FB_NT) {
            if (s_taken >= take) break;                                       // uniform: read after the barrier below
            uint32_t doc[4]; bool m[4]; uint32_t c = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint64_t i = base + (uint64_t)threadIdx.x * 4 + u;
                doc[u] = i < n ? np[sp->rank_asc ? n - 1 - i : i] : 0xFFFFFFFFu;
                m[u] = false;
                if (doc[u] < N && bm_bit(rmap, doc[u])) {
                    const u64 dbw = __hip_atomic_load(&dbmap[doc[u] >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    m[u] = !((dbw >> (doc[u] & 63)) & 1ull);
                }
                c += m[u] ? 1u : 0u;
            }
            uint32_t tot;
            uint32_t p = s_taken + block_excl_scan<FB_NT>(c, &tot, lds);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!m[u]) continue;
                if (p < take) { hits[r0 + p] = doc[u]; atomicOr(&dbmap[doc[u] >> 6], 1ull << (doc[u] & 63)); }
                p++;
            }
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) s_taken = s_taken + tot < take ? s_taken + tot : take;
            __syncthreads();
        }
        __syncthreads();
        if (threadIdx.x == 0) s_pos = r0 + s_taken;
        __syncthreads();
    }
    // 3. every skip-th document of rmap
    bool over = false;
    for (uint32_t base = 0; base < W; base += FB_NT) {
        const uint32_t w = base + threadIdx.x;
        const u64 x = w < W ? rmap[w] : 0ull;
        uint32_t tot;
        const uint32_t ord0 = s_run + block_excl_scan<FB_NT>((uint32_t)__popcll(x), &tot, lds);
        // the bits of this word whose ordinal in rmap is a multiple of skip and that are not in S yet
        u64 sel = 0;
        if (x) {
            const u64 db = __hip_atomic_load(&dbmap[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u64 y = x; uint32_t o = ord0;
            while (y) {
                const int b = __ffsll((long long)y) - 1; y &= y - 1;
                if (o % skip == 0 && !((db >> b) & 1ull)) sel |= 1ull << b;
                o++;
            }
        }
        uint32_t tot2;
        uint32_t off = s_pos + block_excl_scan<FB_NT>((uint32_t)__popcll(sel), &tot2, lds);
        if (s_pos + tot2 > cap) over = true;                                  // uniform
        else if (sel) emit_bits(sel, w << 6, hits, off);
        __syncthreads();
        if (threadIdx.x == 0) { s_run += tot; if (!over) s_pos += tot2; }
        __syncthreads();
        if (over) break;
    }
    if (threadIdx.x == 0) {
        if (over) {                      // the reference would write past its malloc here
            atomicMax(sp->err, (uint32_t)(-MGPU_E_LIMIT));
            *sp->num_hits = 0; *sp->num_rank = 0;
        } else { *sp->num_hits = s_pos; *sp->num_rank = s_pos ? s_pos - 1 : 0; }
    }
}

// after k_fast_build: the tile work list again, over the lists as they are now (samples for the fast-ranked squeries)
__global__ __launch_bounds__(MG_BLOCK) void k_tiles_again(const SqDev *sqs, uint32_t nsq, uint2 *tile_list, uint32_t *tile_count) {
    const uint32_t i = blockIdx.x * MG_BLOCK + threadIdx.x;
    if (i >= nsq) return;
    const SqDev *sp = &sqs[i];
    if (sp->aux) return;
    const uint32_t M = *sp->num_hits;
    const uint32_t ntiles = (M + MG_BLOCK - 1) / MG_BLOCK;
    if (!ntiles) return;
    const uint32_t base = atomicAdd(tile_count, ntiles);
    for (uint32_t t = 0; t < ntiles; t++) tile_list[base + t] = make_uint2(i, t);
}

// get_facet_totalcount (squery.c:398-408) for the facet lists of the fast-ranked squeries: the sampled count of every
// listed id is replaced by |facetid2bmap[id, field] AND docid_map|.  One workgroup per listed id (grid: lists x the
// longest list of the batch): an id costs a binary search of dependent loads and a walk of its posting, and a list has
// up to 2 * maxFacetResults of them -- one workgroup per list took 0.63 ms per batch of the default-mode stream.
struct RecountJob { uint32_t sq; uint32_t field; u64 *out_hi; const u64 *out_lo; const uint32_t *out_n; };
__global__ __launch_bounds__(MG_BLOCK) void k_facet_recount(const SqDev *sqs, const RecountJob *jobs) {
    const RecountJob j = jobs[blockIdx.x];
    const SqDev *sp = &sqs[j.sq];
    if (!*sp->fast_flag) return;
    uint32_t n = *j.out_n;
    if (n == 0xFFFFFFFFu) return;                                             // selection unfinished: the host reports it
    __shared__ uint32_t red[MG_BLOCK / 64];
    const uint32_t f = j.field, nord = sp->sh.fa_nord[f], W = sp->sh.W, Wp = sp->sh.Wp;
    const uint32_t *ord2id = sp->sh.fa_sorted_id[f];      // ids ascending: an id's position = its posting's offset from the field's base pid
    const u64 *D = (const u64 *)sp->docid_map;
    for (uint32_t x = blockIdx.y; x < n; x += gridDim.y) {
        const uint32_t id = (uint32_t)j.out_lo[x];
        uint32_t lo = 0, hi = nord;                                           // the id's place among the field's ids
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (ord2id[mid] < id) lo = mid + 1; else hi = mid; }
        const uint32_t pid = sp->fac_pid_base[f] + lo;
        uint32_t c = 0;
        if (lo < nord && ord2id[lo] == id && pid < sp->sh.num_postings) {
            const int32_t di = sp->sh.post_dense[pid];
            if (di >= 0) {
                const u64 *row = (const u64 *)sp->sh.dense + (size_t)di * Wp;
                for (uint32_t w = threadIdx.x; w < W; w += MG_BLOCK) c += (uint32_t)__popcll(row[w] & D[w]);
            } else {
                const u64 a = sp->sh.post_off[pid], b = sp->sh.post_off[pid + 1];
                for (u64 e = a + threadIdx.x; e < b; e += MG_BLOCK) c += bm_bit(D, sp->sh.post_doc[e]) ? 1u : 0u;
            }
        }
        c = wave_sum32(c);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t t = 0;
            for (int w2 = 0; w2 < MG_BLOCK / 64; w2++) t += red[w2];
            j.out_hi[x] = (u64)(0xFFFFFFFFu - t);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ checked accessors
// MGPU_CHECKED build (make gpu-checked): every global access of the rank / doc-processing kernels whose index is DATA
// (a docid from a hit list, a header word of a record, an ordinal or an offset staged through LDS) is compared with the
// extent the mirror recorded; the first class of violation is reported through the squery's err word
// (0x40000000 | site << 24 | low 24 bits of the index -> MGPU_E_HIP "checked accessor") and the access is NOT made.
// Kernel edits run as a checked build over the whole GPU suite before they run unchecked.  The ordinal guards
// in front of the histogram atomics (sites 4 and 9) are permanent: one compare each.
__device__ __forceinline__ bool mg_ck_always(uint32_t *err, uint32_t site, u64 idx, u64 ext) {
    if (idx < ext) return true;
    atomicMax(err, 0x40000000u | (site << 24) | (uint32_t)(idx & 0xFFFFFFu));
    return false;
}
#ifdef MGPU_CHECKED
#define CK(site, idx, ext) mg_ck_always(sp->err, (site), (u64)(idx), (u64)(ext))
#else
#define CK(site, idx, ext) true
#endif

// ------------------------------------------------------------------ ranking
struct RankOut { uint32_t typos, prox, field, pos, exact; };

// one full-rate 24-bit multiply and a fold (32-bit multiplies are quarter rate and the rank kernel hashes 32 words per
// lane and tile); the host builds the tables with the same function (mgpu_api.hip, wid table)
__device__ __forceinline__ uint32_t hash_wid(uint32_t h) {
    h = __umul24(h, MG_WID_HASH_K);
    return h ^ (h >> 15);
}
__device__ __forceinline__ bool probe_wid(const uint2 *htab, uint32_t hmask, uint32_t wid, uint32_t &val) {
    for (uint32_t h = hash_wid(wid) & hmask;; h = (h + 1) & hmask) {
        const uint2 s = htab[h];
        if (s.x == wid) { val = s.y; return true; }
        if (s.x == MG_WID_EMPTY) return false;
    }
}

// (slot, position) pairs of one document, packed (slot << 24) | (priority << 16) | position.
// Two homes: a lane-strided LDS slice (fast path) or a slice of the global pool (overflow path).
struct LdsPairs {
    uint32_t *base; int cap;
    __device__ __forceinline__ uint32_t get(int j) const { return base[j * MG_BLOCK]; }
    __device__ __forceinline__ void set(int j, uint32_t v) { base[j * MG_BLOCK] = v; }
};
struct PoolPairs {
    uint32_t *base; int cap;
    __device__ __forceinline__ uint32_t get(int j) const { return base[j]; }
    __device__ __forceinline__ void set(int j, uint32_t v) { base[j] = v; }
};

// insertion sort: slot-major, then position
template <class PS>
__device__ __forceinline__ void sort_pairs(PS &p, int n) {
    for (int i = 1; i < n; i++) {
        const uint32_t x = p.get(i);
        int j = i - 1;
        while (j >= 0 && p.get(j) > x) { p.set(j + 1, p.get(j)); j--; }
        p.set(j + 1, x);
    }
}

// the proximity walk of docrank.c:188-204 / 303-319 over two sorted runs
template <class PS>
__device__ __forceinline__ int prox_walk(const PS &p, int a0, int a1, int b0, int b1) {
    int i = a0, j = b0, mindiff = 8;
    while (i < a1 && j < b1) {
        const int pos1 = (int)(p.get(i) & 0xFFFFFFu), pos2 = (int)(p.get(j) & 0xFFFFFFu);
        int diff = pos1 - pos2; if (diff < 0) diff = -diff;
        if (diff != 0 && diff < mindiff) { mindiff = diff; if (mindiff == 1) break; }
        if (pos1 > pos2) j++; else i++;
    }
    return mindiff;
}

// calculate_rank (docrank.c:338-369) split in three: per-document state, one step per document
// word that the query's wid table knows, and the final combination.  Every update commutes
// (min, or, clear-to-zero, a set of pairs sorted later), so the words may arrive in any order.
struct RankState {
    u64 single_best;                    // (dist << 40) | (priority << 16) | position, single-word queries
    u64 typ_lo, typ_hi;                 // per-word typos, one byte each (0xFF = unmatched)
    uint32_t exact_mask;
    uint32_t prox1;                     // bit x: word x matched through a concatenated (odd) term
    uint32_t bestpos;
    int np;
    bool comb;                          // the all-words term matched
    bool ovf;                           // the pair store overflowed
};
__device__ __forceinline__ void rank_init(RankState &s) {
    s.single_best = ~0ull; s.typ_lo = ~0ull; s.typ_hi = ~0ull; s.exact_mask = 0; s.prox1 = 0;
    s.bestpos = 0xFFFFFFFFu; s.np = 0; s.comb = false; s.ovf = false;
}

// which proximity slots the positions of a word matched through term `mterm` feed
// (rank_three_terms docrank.c:125-209, rank_many_terms docrank.c:218-328; term_to_word_idx's
// quirk reduces to term / 2)
__device__ __forceinline__ void rank_slots(int nw, int nt, uint32_t mterm, int &slot_a, int &slot_b, bool &keep_a) {
    slot_b = -1; keep_a = true;
    if (nw == 2) { slot_a = (int)mterm; keep_a = slot_a != 2; }     // the pair term only feeds best_position
    else {
        slot_a = (int)mterm >> 1;
        if ((mterm & 1u) && (int)mterm != nt - 1) slot_b = slot_a + 1;
    }
}
__device__ __forceinline__ uint32_t info_first_inline(uint32_t info) { return (((info >> 1) & 0xFFu) << 16) + (info >> 9); }

// the order-free reductions of one matched word; val = dist | mterm << 8 | exact_mask << 16 (wid
// table), first = (priority << 16) + position of its first occurrence
__device__ __forceinline__ void rank_reduce(RankState &s, int nw, int nt, uint32_t val, uint32_t first) {
    s.exact_mask |= val >> 16;
    const uint32_t dist = val & 0xFF, mterm = (val >> 8) & 0xFF;
    if (dist == 0xFF) return;
    if (nw == 1) {
        // rank_single_term (docrank.c:63-112): least distance, then least (field, position)
        const u64 k = ((u64)dist << 40) | (u64)first;
        if (k < s.single_best) s.single_best = k;
        return;
    }
    if (nw == 2) {
        if (mterm == 2) s.comb = true;
        else {
            const int sh8 = (int)mterm * 8;
            if (dist < ((s.typ_lo >> sh8) & 0xFF)) s.typ_lo = (s.typ_lo & ~(0xFFull << sh8)) | ((u64)dist << sh8);
        }
    } else {
        if ((int)mterm == nt - 1) s.comb = true;
        const int widx = (int)mterm >> 1;
        {
            u64 &tw = widx < 8 ? s.typ_lo : s.typ_hi;
            const int sh8 = (widx & 7) * 8;
            if (dist < ((tw >> sh8) & 0xFF)) tw = (tw & ~(0xFFull << sh8)) | ((u64)dist << sh8);
        }
        if (mterm & 1u) {
            s.prox1 |= 1u << widx;
            if ((int)mterm != nt - 1) {
                const int w1 = widx + 1;
                u64 &tw = w1 < 8 ? s.typ_lo : s.typ_hi;
                tw &= ~(0xFFull << ((w1 & 7) * 8));
            }
        }
    }
    if (first < s.bestpos) s.bestpos = first;          // runs are sorted: the first entry is the least
}

// serial form used by the overflow kernel: reductions + the word's (slot, position) pairs.
// info = the docword's position descriptor; pp_doc = the document's slice of the position pool
template <class PS>
__device__ __forceinline__ void rank_consume(RankState &s, int nw, int nt, uint32_t val, uint32_t info,
                                             const uint32_t *pp_doc, PS &ps) {
    const bool inl = info & 1u;
    const uint32_t cnt = inl ? 1u : ((info >> 1) & 0xFFu);
    const uint32_t poff = info >> 9;
    const bool ranked = (val & 0xFF) != 0xFF;
    const uint32_t first = !ranked ? 0u : inl ? info_first_inline(info) : pp_doc[poff];
    rank_reduce(s, nw, nt, val, first);
    if (!ranked || nw == 1) return;
    int slot_a, slot_b; bool keep_a;
    rank_slots(nw, nt, (val >> 8) & 0xFF, slot_a, slot_b, keep_a);
    for (uint32_t j = 0; j < cnt; j++) {
        const uint32_t p = (j == 0) ? first : pp_doc[poff + j];
        if (keep_a) { if (s.np < ps.cap) ps.set(s.np++, ((uint32_t)slot_a << 24) | p); else s.ovf = true; }
        if (slot_b >= 0) { if (s.np < ps.cap) ps.set(s.np++, ((uint32_t)slot_b << 24) | p); else s.ovf = true; }
    }
}

template <class PS>
__device__ __forceinline__ void rank_finish(RankState &s, int nw, uint32_t exact_present, PS &ps, RankOut &ro) {
    const uint32_t exact_mask = s.exact_mask & exact_present;
    ro.exact = (uint32_t)__popc(exact_mask);
    if (nw == 1) {
        ro.prox = 0;
        const bool ex = exact_mask & 1u;
        uint32_t d = (uint32_t)(s.single_best >> 40);
        if (s.single_best == ~0ull) d = 0xFF;
        if (ex) {
            ro.typos = 0;
            if (d == 0) { ro.field = (uint32_t)(s.single_best >> 16) & 0xFF; ro.pos = (uint32_t)s.single_best & 0xFFFF; }
            else { ro.field = 0xFF; ro.pos = 0xFFFF; }
        } else if (d <= 0xFE) {
            ro.typos = d; ro.field = (uint32_t)(s.single_best >> 16) & 0xFF; ro.pos = (uint32_t)s.single_best & 0xFFFF;
        } else { ro.typos = 0xFF; ro.field = 0xFF; ro.pos = 0xFFFF; }
        return;
    }
    // typos: exact words and the all-words term zero them; the sum wraps like the reference's uint8_t
    uint32_t tsum = 0;
    for (int i = 0; i < nw; i++) {
        uint32_t t = (uint32_t)(((i < 8 ? s.typ_lo : s.typ_hi) >> ((i & 7) * 8)) & 0xFF);
        if (s.comb || ((exact_mask >> i) & 1u)) t = 0;
        tsum += t;
    }
    ro.typos = tsum & 0xFF;
    ro.field = (s.bestpos >> 16) & 0xFF;
    ro.pos = s.bestpos & 0xFFFF;
    if (s.comb) { ro.prox = (nw == 2) ? 1u : (uint32_t)(nw - 1); return; }
    const int np = s.np;
    sort_pairs(ps, np);
    uint32_t prox = 0;
    int a0 = 0, a1 = 0;
    while (a1 < np && (ps.get(a1) >> 24) == 0) a1++;
    for (int x = 0; x < nw - 1; x++) {
        const int b0 = a1; int b1 = b0;
        while (b1 < np && (int)(ps.get(b1) >> 24) == x + 1) b1++;
        if (nw > 2 && ((s.prox1 >> x) & 1u)) prox += 1;
        else prox += (uint32_t)prox_walk(ps, a0, a1, b0, b1);
        a0 = b0; a1 = b1;
    }
    ro.prox = prox & 0xFFFF;
}

__device__ __forceinline__ u64 f64_order_key(double x) {
    u64 u = (u64)__double_as_longlong(x);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

// the key layout as 8 packed words {kind, bits, shift, hi}, so that it can live in registers
struct KeyLayout { uint32_t f[8]; int n; };
__device__ __forceinline__ KeyLayout load_key_layout(const SqDev *sp) {
    KeyLayout k;
    const uint32_t *w = (const uint32_t *)sp->kf;
#pragma unroll
    for (int i = 0; i < 8; i++) k.f[i] = w[i];
    k.n = sp->nkf;
    return k;
}
__device__ __forceinline__ void build_key(const KeyLayout &kl, const RankOut &ro, double cmp, u64 &hi, u64 &lo) {
    u128 key = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (i >= kl.n) break;
        const int kind = (int)(int8_t)(kl.f[i] & 0xFF);
        const uint32_t shift = (kl.f[i] >> 16) & 0xFF;
        u64 v;
        switch (kind) {
            case KF_TYPO: v = ro.typos; break;
            case KF_PROX: v = ro.prox; break;
            case KF_POS: v = ro.pos; break;
            case KF_EXACT: v = 255u - ro.exact; break;
            case KF_FIELD: v = ro.field; break;
            case KF_COMP_DESC: v = ~f64_order_key(cmp); break;
            default: v = f64_order_key(cmp); break;
        }
        key |= (u128)v << shift;
    }
    hi = (u64)(key >> 64); lo = (u64)key;
}

__device__ __forceinline__ void or_key_bits(u64 *key_andor, u64 a, u64 b, u64 na, u64 nb) {
    // bits that differ between keys: OR of keys and OR of complements
    a = wave_or64(a); b = wave_or64(b); na = wave_or64(na); nb = wave_or64(nb);
    if ((threadIdx.x & 63) == 0 && (a | b | na | nb)) {
        atomicOr((u64 *)&key_andor[0], na); atomicOr((u64 *)&key_andor[1], nb);
        atomicOr((u64 *)&key_andor[2], a);  atomicOr((u64 *)&key_andor[3], b);
    }
}

// perform_doc_rank (docrank.c:414-443) for every hit, one tile of 256 hits per work item.
// Phase 1 streams the docword runs of the tile: each group of 16 lanes takes one document at a
// time and reads its run as one coalesced 128-byte load, keeping only the words the query
// knows -- a handful per document -- as records in LDS.  Phase 2 ranks one document per lane from
// its records.  Documents with more matches or positions than the LDS slices hold go to the
// overflow list and are ranked by k_rank_overflow.
// appends `v` to list[*n] for every lane whose `pred` is set, with one atomic per wave
__device__ __forceinline__ void wave_append(bool pred, uint32_t v, uint32_t *list, uint32_t *n) {
    const u64 m = __ballot(pred);
    if (!m) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    uint32_t at = 0;
    if (lane == leader) at = atomicAdd(n, (uint32_t)__popcll(m));
    at = __shfl(at, leader, 64);
    if (pred) list[at + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = v;
}

extern __shared__ uint2 dyn_tab[];
// LEN: every shard of the batch has docids below 2^24, so a record's length (ShardDev::rec_len, 16-byte units) rides in
// the top byte of the tile's docid words and the lanes beyond it load nothing
template <bool LEN>
__global__ __launch_bounds__(MG_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_rank_tile(const SqDev *sqs, const uint32_t *blob,
                                                        const uint2 *tile_list, const uint32_t *tile_count) {
    __shared__ uint32_t s_doc[MG_BLOCK], s_pbase[MG_BLOCK], s_cnt[MG_BLOCK], s_np[MG_BLOCK];
    __shared__ double s_val[MG_BLOCK];
    __shared__ uint32_t s_nord[MGPU_MAX_FIELDS];      // ordinals per facet field of the squery in hand (the ordinal guard's extents)
    __shared__ uint2 s_match[MG_BLOCK * MG_MCAP];
    __shared__ uint32_t s_pairs[MG_BLOCK * (MG_PCAP + 1)];         // + a spare row that takes the staging's out-of-range words
    const uint32_t ntiles = *tile_count;
    const int skip = mg_dbg_skip;
    const uint32_t tid = threadIdx.x;
    const uint32_t grp = tid >> 4, gl = tid & 15;
    // a workgroup takes a run of consecutive tiles: a squery's tiles sit together on the list, so its wid table is
    // staged into LDS once per run instead of once per tile
    const uint32_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    const uint32_t w_end = (blockIdx.x + 1) * per < ntiles ? (blockIdx.x + 1) * per : ntiles;
    uint32_t tab_of = 0xFFFFFFFFu;                   // squery whose table dyn_tab holds
    uint32_t nord_of = 0xFFFFFFFFu;                  // squery whose field sizes s_nord holds
    for (uint32_t w = blockIdx.x * per; w < w_end; w++) {
        DBG_T(t0);
        // the work item is the same for every lane: say so, so that the descriptor is read with scalar loads
        const uint2 item_v = tile_list[w];
        const uint2 item = make_uint2(__builtin_amdgcn_readfirstlane(item_v.x), __builtin_amdgcn_readfirstlane(item_v.y));
        // Every field of the descriptor that this tile needs, read in ONE burst of independent loads
        // (the compiler cannot prove the descriptor constant across the stores below and would
        // otherwise re-read each field, serially, at its point of use).
        const SqDev *sp = &sqs[item.x];
        const int do_rank = sp->do_rank;
        const uint32_t *p_num_hits = sp->num_hits;
        const uint32_t *p_hits = sp->hits;
        const uint32_t hmask = sp->hash_mask, hash_off = sp->hash_off;
        const int rb = sp->rank_by;
        const uint32_t rec_nnum = sp->sh.rec_nnum, rec_stride = sp->sh.rec_stride;
        const uint4 *rec = (const uint4 *)sp->sh.rec;
        const uint8_t *rec_len = sp->sh.rec_len;
        const uint2 *dw = sp->sh.dw;
        const int nw = sp->num_words, nt = sp->num_terms;
        const uint32_t exact_present = sp->exact_present;
        u64 *p_key_hi = (u64 *)sp->key_hi, *p_key_lo = (u64 *)sp->key_lo, *p_key_andor = (u64 *)sp->key_andor;
        uint32_t *p_ovf_cnt = sp->ovf_cnt, *p_ovf_list = sp->ovf_list;
        const KeyLayout kl = load_key_layout(sp);
        const double *col = sp->sh.num_col[rb >= 0 ? rb : 0];
        const int fac_here = sp->fac_in_rank;
        const uint32_t fac_mask = sp->hist_mask, fdirect = sp->fdirect_mask, fhash = sp->fhash_mask;
        const int nfac = sp->sh.num_facets;
#ifdef MGPU_CHECKED
        const uint32_t x_N = sp->sh.N, x_dw = sp->sh.dw_len, x_cap = sp->hit_cap, x_rec = sp->sh.rec_words;
#endif
        if (do_rank < 0) continue;                   // no hits requested: nobody reads the keys
        const uint32_t M = *p_num_hits;
        const uint2 *gtab = (const uint2 *)(blob + hash_off);
        const uint32_t t = item.y * MG_BLOCK + tid;
        const bool valid = t < M;
        const uint32_t docid = valid ? p_hits[t] : 0xFFFFFFFFu;
        const bool val_in_rec = rb >= 0 && (uint32_t)rb < rec_nnum;
        if (do_rank || val_in_rec) {
            if (do_rank && hmask < MG_LDS_HASH && tab_of != item.x) {
                for (uint32_t i = tid; i <= hmask; i += MG_BLOCK) dyn_tab[i] = gtab[i];
                tab_of = item.x;
            }
            if (fac_here && nord_of != item.x) {           // once per run of tiles, with the loads of the tile's first step
                if (tid < MGPU_MAX_FIELDS) s_nord[tid] = sp->sh.fa_nord[tid];
                nord_of = item.x;
            }
            {
                uint32_t dl = docid;
                if (LEN && valid && CK(1, docid, x_N)) dl = docid | ((uint32_t)rec_len[docid] << 24);
                s_doc[tid] = dl; s_cnt[tid] = 0; s_np[tid] = 0;
            }
            __syncthreads();
            DBG_T(t1); DBG_ADD(0, t0, t1);
            const uint32_t S16 = rec_stride >> 4;                  // uint4 per record
            const uint32_t dmask = LEN ? 0xFFFFFFu : 0xFFFFFFFFu;
            // group g owns documents 16g .. 16g+15 of the tile and reads each record as ONE coalesced
            // load of 16 bytes per lane (256 B); eight records are in flight per lane at a time.
            // two explicit probe paths: through a runtime-selected pointer the compiler would emit
            // flat loads (LDS or global decided per access) and wait on each one
            const bool tab_in_lds = do_rank && hmask < MG_LDS_HASH;
            // facet staging (phase 1): the rows of s_pairs that this lane's four record words go to (the spare row for
            // words outside the section's first MG_PCAP)
            const int fx0 = 4 * (int)gl - (int)(4 + 2 * rec_nnum);
            const bool stage_lane = fx0 > -4 && fx0 < MG_PCAP;
            const uint32_t srow0 = (uint32_t)((fx0 >= 0 && fx0 < MG_PCAP) ? fx0 : MG_PCAP) * MG_BLOCK;
            const uint32_t srow1 = (uint32_t)((fx0 + 1 >= 0 && fx0 + 1 < MG_PCAP) ? fx0 + 1 : MG_PCAP) * MG_BLOCK;
            const uint32_t srow2 = (uint32_t)((fx0 + 2 >= 0 && fx0 + 2 < MG_PCAP) ? fx0 + 2 : MG_PCAP) * MG_BLOCK;
            const uint32_t srow3 = (uint32_t)((fx0 + 3 >= 0 && fx0 + 3 < MG_PCAP) ? fx0 + 3 : MG_PCAP) * MG_BLOCK;
            const uint32_t gshift = tid & 48u;                     // the group's first bit in a wave ballot
            const uint32_t gbelow = (1u << gl) - 1u;               // the group's lanes below this one
            // match records are r-major (record r of document d at [r * MG_BLOCK + d]): phase 2 reads its document's
            // records lane-strided, without bank conflicts
            auto record = [&](uint32_t d, uint32_t val, uint32_t info) {
                const uint32_t r = atomicAdd(&s_cnt[d], 1u);
                if (r < MG_MCAP) s_match[r * MG_BLOCK + d] = make_uint2(val, info);
            };
            auto take = [&](uint32_t d, uint32_t wid, uint32_t info) {          // full probe (tails, global tables)
                uint32_t val;
                const bool found = tab_in_lds ? probe_wid(dyn_tab, hmask, wid, val) : probe_wid(gtab, hmask, wid, val);
                if (found) record(d, val, info);
            };
            // after a first probe that met another key: continue from the next slot
            auto take_from = [&](uint32_t d, uint32_t wid, uint32_t info, uint32_t h) {
                for (h = (h + 1) & hmask;; h = (h + 1) & hmask) {
                    const uint2 e = dyn_tab[h];
                    if (e.x == wid) { record(d, e.y, info); return; }
                    if (e.x == MG_WID_EMPTY) return;
                }
            };
            for (int half = 0; half < 2; half++) {
                uint4 v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t dl = s_doc[grp * 16 + half * 8 + k];
                    const uint32_t doc = dl & dmask, len16 = LEN ? (dl >> 24) : S16;
                    v[k] = (dl != 0xFFFFFFFFu && gl < len16 && gl < S16 && CK(1, doc, x_N)) ? rec[(size_t)doc * S16 + gl] : make_uint4(0u, 0u, 0u, 0u);
                }
                // headers (lane 0 of the group holds them), rank value, and which of the lane's two 8-byte
                // slots are docword entries
                uint32_t in_a = 0, in_b = 0, tail = 0;
                const uint32_t dbase = grp * 16 + half * 8;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t h0 = __shfl(v[k].x, 0, 16), h3 = __shfl(v[k].w, 0, 16);
                    const uint32_t nin = h0 >> 16, ws = h3 >> 16;
                    const uint32_t s0 = 2 * gl;
                    // ws <= slot < ws + nin as one unsigned compare; no branches in this loop
                    in_a |= (s0 - ws < nin ? 1u : 0u) << k;
                    in_b |= (s0 + 1u - ws < nin ? 1u : 0u) << k;
                    tail |= ((nin + ws > 32 || (h0 & 0xFFFFu) > nin) ? 1u : 0u) << k;
                }
                // What single lanes of the group copy out of the record runs as ONE predicated region per role with the
                // eight documents inside (the roles do not depend on k), instead of a handful of regions per document:
                // exec-mask bookkeeping is most of this kernel's scalar instructions.
                if (gl == 0) {
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        s_pbase[dbase + k] = v[k].y;
                        if (fac_here) s_np[dbase + k] = (v[k].w >> 8) & 1u;       // "the facet section is in the record"
                    }
                }
                if (fac_here && stage_lane) {
                    // the record's facet section (from word 4 + 2 * rec_nnum) is staged in the pair slices, which are
                    // idle until phase 2
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        s_pairs[srow0 + dbase + k] = v[k].x; s_pairs[srow1 + dbase + k] = v[k].y;
                        s_pairs[srow2 + dbase + k] = v[k].z; s_pairs[srow3 + dbase + k] = v[k].w;
                    }
                }
                if (val_in_rec && gl == (uint32_t)(1 + (rb >> 1))) {
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const u64 bits = (rb & 1) ? (((u64)v[k].w << 32) | v[k].z) : (((u64)v[k].y << 32) | v[k].x);
                        s_val[dbase + k] = __longlong_as_double((long long)bits);
                    }
                }
                if (do_rank && !(skip & 4)) {
                    if (tab_in_lds) {
                        // first probes of all sixteen entries issued together; the table is sparse, so they
                        // settle nearly every word (empty slot = not a query word)
#pragma unroll
                        for (int q4 = 0; q4 < 8; q4 += 4) {
                            uint2 pa[4], pb[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                pa[k] = dyn_tab[hash_wid(v[q4 + k].x) & hmask];
                                pb[k] = dyn_tab[hash_wid(v[q4 + k].z) & hmask];
                            }
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const uint32_t d = grp * 16 + half * 8 + q4 + k;
                                const uint4 e = v[q4 + k];
                                // The sixteen lanes of the group hold the document's 32 entries: the record slot of
                                // a match is its rank among the group's matches (two ballots), no LDS atomic and no
                                // wait for one.  The document belongs to this group alone, so the plain store of its
                                // count precedes (wave program order) the atomics of the rare paths below.
                                const bool use_a = (in_a >> (q4 + k)) & 1u, use_b = (in_b >> (q4 + k)) & 1u;
                                const bool ma = use_a && pa[k].x == e.x, mb = use_b && pb[k].x == e.z;
                                const uint32_t ga = (uint32_t)(__ballot(ma) >> gshift) & 0xFFFFu;
                                const uint32_t gb = (uint32_t)(__ballot(mb) >> gshift) & 0xFFFFu;
                                const uint32_t ra = (uint32_t)__popc(ga & gbelow) + (uint32_t)__popc(gb & gbelow);
                                const uint32_t rb2 = ra + (ma ? 1u : 0u);
                                if (ma && ra < MG_MCAP) s_match[ra * MG_BLOCK + d] = make_uint2(pa[k].y, e.y);
                                if (mb && rb2 < MG_MCAP) s_match[rb2 * MG_BLOCK + d] = make_uint2(pb[k].y, e.w);
                                if (gl == 0) s_cnt[d] = (uint32_t)__popc(ga) + (uint32_t)__popc(gb);
                                if (use_a && !ma && pa[k].x != MG_WID_EMPTY) take_from(d, e.x, e.y, hash_wid(e.x) & hmask);
                                if (use_b && !mb && pb[k].x != MG_WID_EMPTY) take_from(d, e.z, e.w, hash_wid(e.z) & hmask);
                            }
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const uint32_t d = grp * 16 + half * 8 + k;
                            if ((in_a >> k) & 1u) take(d, v[k].x, v[k].y);
                            if ((in_b >> k) & 1u) take(d, v[k].z, v[k].w);
                        }
                    }
                    // records longer than 256 B, then the words that did not fit the record at all (rare)
                    if (__any(tail != 0)) {
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            if (!((tail >> k) & 1u)) continue;
                            const uint32_t d = grp * 16 + half * 8 + k;
                            const uint32_t doc = s_doc[d] & dmask;
                            const uint32_t h0 = __shfl(v[k].x, 0, 16), h2 = __shfl(v[k].z, 0, 16), h3 = __shfl(v[k].w, 0, 16);
                            const uint32_t nin = h0 >> 16, ws = h3 >> 16, nwords_k = h0 & 0xFFFFu;
                            for (uint32_t c = 16 + gl; c < S16; c += 16) {
                                if (!CK(1, doc, x_N)) break;
                                const uint4 x = rec[(size_t)doc * S16 + c];
                                if (2 * c >= ws && 2 * c < ws + nin) take(d, x.x, x.y);
                                if (2 * c + 1 >= ws && 2 * c + 1 < ws + nin) take(d, x.z, x.w);
                            }
                            for (uint32_t j = nin + gl; j < nwords_k; j += 16) {
                                if (!CK(2, (u64)h2 + (j - nin), x_dw)) break;
                                const uint2 w2 = dw[h2 + (j - nin)];
                                take(d, w2.x, w2.y);
                            }
                        }
                    }
                }
            }
            __syncthreads();
            DBG_T(t2); DBG_ADD(1, t1, t2);
            // Phase 1c, one lane per document: the hit's facet ordinals (process_facet_result, docrank.c:371-389)
            // go from the staged section into fac_ord, so that k_docproc counts them from a coalesced array
            // instead of visiting the same record a second time.  Anything but "one value, staged" is left to
            // k_docproc's own lookup.
            if (fac_here) {
                if (valid) {
                    const bool in_rec = s_np[tid] & 1u;
                    const uint32_t hist_mask = fac_mask;
                    uint32_t pos = 0;
                    bool staged_ok = in_rec;
                    for (int f = 0; f < nfac; f++) {
                        uint32_t code = MG_FAC_LOOKUP;
                        if (staged_ok && pos < MG_PCAP) {
                            const uint32_t cnt = s_pairs[pos * MG_BLOCK + tid];
                            if (cnt == 0) code = MG_FAC_NONE;
                            else if (cnt == 1 && pos + 1 < MG_PCAP) code = s_pairs[(pos + 1) * MG_BLOCK + tid];
                            pos += 1 + cnt;
                        } else staged_ok = false;
                        // the ordinal came through LDS staging and indexes histograms and ordinal -> id tables from here
                        // on: one that is not an ordinal of the field is reported, never used (permanent guard)
                        if (code < MG_FAC_LOOKUP && code >= s_nord[f]) { mg_ck_always(sp->err, 4, code, s_nord[f]); code = MG_FAC_NONE; }
                        if (((hist_mask >> f) & 1u) && CK(5, t, x_cap)) sp->fac_ord[f][t] = code;
                        // directly counted field: one fire-and-forget atomic, nothing waits for it
                        // (unless the select stage counts the field by hashing the ordinals: few enough hits)
                        if (((fdirect >> f) & 1u) && code < MG_FAC_LOOKUP && !(((fhash >> f) & 1u) && M <= MG_FHASH_MAX_M))
                            atomicAdd(&sp->hist[f][code], 1u);
                    }
                }
            }
        }
        DBG_T(t3); DBG_ADD(16, t0, t3);
        u64 a = 0, b = 0, na = 0, nb = 0;
        if (valid) {
            RankOut ro = {0, 0, 0, 0, 0};
            bool ok = true;
            if (do_rank && !(skip & 2)) {
                // Phase 2, one lane per document, all in the lane's own LDS columns (no barrier since phase 1's):
                // the word positions -- the only dependent global reads of the ranking -- are fetched for all of the
                // document's records in one batch of independent loads (first three positions of every word), then the
                // records are reduced and their (slot, position) pairs laid down, sorted and walked.
                const uint32_t cnt = s_cnt[tid];
                if (cnt > MG_MCAP) ok = false;
                else {
                    // the document's positions: one index into the records-then-pool array, wherever they lie
                    const uint32_t *pdoc = (const uint32_t *)rec + s_pbase[tid];
#ifdef MGPU_CHECKED
                    const u64 x_pbase = s_pbase[tid], x_pext = x_rec;
#endif
                    uint32_t p0[MG_MCAP], p1[MG_MCAP], p2[MG_MCAP];
#pragma unroll
                    for (int r = 0; r < MG_MCAP; r++) { p0[r] = 0; p1[r] = 0; p2[r] = 0; }
#pragma unroll
                    for (int r = 0; r < MG_MCAP; r++) {
                        if (!__any((uint32_t)r < cnt)) break;                 // nobody in the wave has this many records
                        if ((uint32_t)r < cnt) {
                            const uint2 m = s_match[r * MG_BLOCK + tid];
                            const uint32_t info = m.y;
                            if ((m.x & 0xFF) != 0xFF) {                       // exact-only entries need no positions
                                if (info & 1u) p0[r] = info_first_inline(info);
                                else {
                                    uint32_t pc = (info >> 1) & 0xFFu;
                                    const uint32_t *pd = pdoc + (info >> 9);
                                    if (pc > 0 && !CK(3, x_pbase + (info >> 9) + pc - 1, x_pext)) pc = 0;
                                    if (pc > 0) p0[r] = pd[0];
                                    if (nw > 1 && pc > 1) p1[r] = pd[1];
                                    if (nw > 1 && pc > 2) p2[r] = pd[2];
                                }
                            }
                        }
                    }
                    RankState st; rank_init(st);
                    LdsPairs ps = {&s_pairs[tid], MG_PCAP};
#pragma unroll
                    for (int r = 0; r < MG_MCAP; r++) {
                        if (!__any((uint32_t)r < cnt)) break;
                        if ((uint32_t)r >= cnt) continue;
                        const uint2 m = s_match[r * MG_BLOCK + tid];
                        rank_reduce(st, nw, nt, m.x, p0[r]);
                        if ((m.x & 0xFF) == 0xFF || nw == 1) continue;
                        const uint32_t info = m.y;
                        const uint32_t pc = (info & 1u) ? 1u : ((info >> 1) & 0xFFu);
                        int slot_a, slot_b; bool keep_a;
                        rank_slots(nw, nt, (m.x >> 8) & 0xFF, slot_a, slot_b, keep_a);
                        const uint32_t per = (keep_a ? 1u : 0u) + (slot_b >= 0 ? 1u : 0u);
                        if (!per) continue;
                        if ((uint32_t)st.np + per * pc > MG_PCAP) { st.ovf = true; continue; }
                        // Both pairs of a position are stored unconditionally and the cursor moves past the ones that count: a
                        // store that does not count is overwritten by the next one, and the last one lands at most in the
                        // spare row (np + per * pc <= MG_PCAP was checked above, so np never passes MG_PCAP).  A record with
                        // no position at all (pc == 0: no writer makes one) stores nothing.
                        const uint32_t sa = (uint32_t)slot_a << 24, sb = (uint32_t)slot_b << 24;
                        const int ia = keep_a ? 1 : 0, ib = slot_b >= 0 ? 1 : 0;
                        int np = st.np;
                        if (pc > 0) { ps.set(np, sa | p0[r]); np += ia; ps.set(np, sb | p0[r]); np += ib; }
                        if (pc > 1) { ps.set(np, sa | p1[r]); np += ia; ps.set(np, sb | p1[r]); np += ib; }
                        if (pc > 2) { ps.set(np, sa | p2[r]); np += ia; ps.set(np, sb | p2[r]); np += ib; }
                        if (pc > 3) {
                            const uint32_t *pd = pdoc + (info >> 9);
                            for (uint32_t j = 3; j < pc; j++) {
                                if (!CK(3, x_pbase + (info >> 9) + j, x_pext)) break;
                                const uint32_t pv = pd[j];
                                ps.set(np, sa | pv); np += ia; ps.set(np, sb | pv); np += ib;
                            }
                        }
                        st.np = np;
                    }
                    if (st.ovf) ok = false; else rank_finish(st, nw, exact_present, ps, ro);
                }
                if (!ok) { const uint32_t oi = atomicAdd(p_ovf_cnt, 1u); if (CK(6, oi, x_cap)) p_ovf_list[oi] = t; }
            }
            if (ok) {
                u64 hi, lo;
                build_key(kl, ro, val_in_rec ? s_val[tid] : ((rb >= 0 && CK(7, docid, x_N)) ? col[docid] : 0.0), hi, lo);
                if (CK(8, t, x_cap)) { p_key_hi[t] = hi; p_key_lo[t] = lo; }
                a = hi; b = lo; na = ~hi; nb = ~lo;
            }
        }
        or_key_bits(p_key_andor, a, b, na, nb);
        __syncthreads();                             // LDS is reused by the next work item
        DBG_T(t4); DBG_ADD(2, t3, t4); DBG_CNT(3);
    }
}

// The documents the tile kernel gave up on (long position lists): one document per lane, walking
// its docwords serially with the pair store in the context's global pool.
__global__ __launch_bounds__(MG_BLOCK) void k_rank_overflow(const SqDev *sqs, const uint32_t *blob,
                                                            u64 *pool, uint32_t *pool_used, uint64_t pool_cap) {
    const SqDev &sq = sqs[blockIdx.y];
    if (sq.do_rank <= 0) return;
    const uint32_t M = *sq.ovf_cnt;
    if (blockIdx.x * MG_BLOCK >= M) return;
    const uint2 *htab = (const uint2 *)(blob + sq.hash_off);
    const uint32_t hmask = sq.hash_mask;
    const int nw = sq.num_words, nt = sq.num_terms;
    const KeyLayout kl = load_key_layout(&sq);
    u64 a = 0, b = 0, na = 0, nb = 0;
    for (uint32_t base = blockIdx.x * MG_BLOCK; base < M; base += gridDim.x * MG_BLOCK) {
        const uint32_t t = base + threadIdx.x;
        if (t >= M) continue;
        const uint32_t hidx = sq.ovf_list[t];
        const uint32_t docid = sq.hits[hidx];
        const uint32_t e0 = sq.sh.dw_off[docid], e1 = sq.sh.dw_off[docid + 1];
        const uint32_t pbase = sq.sh.pp_off[docid];
        // bound of the pairs a document can produce: twice its token count
        const uint32_t ntok = (e1 - e0) + (sq.sh.pp_off[docid + 1] - pbase);
        const uint32_t need = ntok + 1;                  // u64 units = 2 packed pairs each
        const uint32_t at = atomicAdd(pool_used, need);
        if ((uint64_t)at + need > pool_cap) { atomicMax(sq.err, (uint32_t)(-MGPU_E_SCRATCH)); continue; }
        RankState st; rank_init(st);
        PoolPairs ps = {(uint32_t *)(pool + at), (int)(2 * need)};
        for (uint32_t e = e0; e < e1; e++) {
            const uint2 wi = sq.sh.dw[e];
            uint32_t val;
            if (probe_wid(htab, hmask, wi.x, val)) rank_consume(st, nw, nt, val, wi.y, sq.sh.pp + pbase, ps);
        }
        RankOut ro = {0, 0, 0, 0, 0};
        rank_finish(st, nw, sq.exact_present, ps, ro);
        u64 hi, lo;
        build_key(kl, ro, (sq.rank_by >= 0) ? sq.sh.num_col[sq.rank_by][docid] : 0.0, hi, lo);
        sq.key_hi[hidx] = hi; sq.key_lo[hidx] = lo;
        a |= hi; b |= lo; na |= ~hi; nb |= ~lo;
    }
    or_key_bits((u64 *)sq.key_andor, a, b, na, nb);
}


// ------------------------------------------------------------------ facets + aggregations
// process_facet_result (docrank.c:371-389) and consume_*_agg (metric-aggs.c) over the hit list.
// Facet ids were turned into dense ordinals by the mirror, so the hashtable of hashtable.c
// becomes a histogram: privatised in LDS when the field has <= MG_LDS_HIST values.
// pre: 1 = the pass BEFORE a fast-ranked squery's hit list is replaced by its sample: aggregations only, and only for
// squeries whose fast_flag is set (calculate_agg over all matching documents, docrank.c:553-557); 0 = the normal pass,
// in which those squeries count facets only (perform_doc_processing, docrank.c:409).
#define DP_ITEMS 256                         // work items a workgroup looks at at a time
#define DP_AGGS 2                            // aggregations reduced together per pass over a slice
__global__ __launch_bounds__(MG_BLOCK) void k_docproc(const SqDev *sqs, const uint2 *tile_list, const uint32_t *tile_count, uint32_t lds_bins, int pre) {
    uint32_t *hist = (uint32_t *)dyn_tab;        // lds_bins words of dynamic LDS: the largest field of the batch that is privatised
    __shared__ AggPart red[DP_AGGS][MG_BLOCK / 64];
    __shared__ uint2 s_items[DP_ITEMS];
    const uint32_t ntiles = *tile_count;
    // A workgroup takes a run of consecutive work items: a squery's items sit together on the list (k_order), so the
    // privatised histogram of a field is zeroed and flushed once per squery and run instead of once per 256 hits -- the
    // flush is one returning global atomic per non-empty bin (first touch is remembered), which was most of this kernel.
    const uint32_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    const uint32_t run_end = (blockIdx.x + 1) * per < ntiles ? (blockIdx.x + 1) * per : ntiles;
    for (uint32_t c0 = blockIdx.x * per; c0 < run_end; c0 += DP_ITEMS) {
        const uint32_t nc = run_end - c0 < DP_ITEMS ? run_end - c0 : DP_ITEMS;
        __syncthreads();
        if (threadIdx.x < nc) s_items[threadIdx.x] = tile_list[c0 + threadIdx.x];
        __syncthreads();
        uint32_t x0 = 0;
        while (x0 < nc) {
            // [x0, x1): the items of one squery
            const uint32_t sqi = __builtin_amdgcn_readfirstlane(s_items[x0].x);
            uint32_t x1 = x0 + 1;
            while (x1 < nc && __builtin_amdgcn_readfirstlane(s_items[x1].x) == sqi) x1++;
            const SqDev &sq = sqs[sqi];
            // The hit list is cut into nproc_blocks interleaved slices (fixed by the shard size so that
            // the f64 reduction tree never depends on the batch); work item (sq, vb) is slice vb, and
            // tiles beyond the slice count have nothing to add.
            const uint32_t nvb = sq.nproc_blocks;
            const bool fast = sq.fast_eligible && *sq.fast_flag;
            const bool skip_all = (sq.hist_mask == 0 && sq.num_aggs == 0) || (pre && !(fast && sq.num_aggs));
            if (skip_all) { x0 = x1; continue; }
            const bool have_ords = sq.fac_in_rank != 0;
            const uint32_t M = *sq.num_hits;
            const uint32_t stride = nvb * MG_BLOCK;
            const uint32_t S4 = sq.sh.rec_stride >> 2;
            const uint32_t *rec = sq.sh.rec;
            const uint32_t nnum_in = sq.sh.rec_nnum;
            const uint32_t *hits = sq.hits;
            const int nfac = sq.sh.num_facets, naggs = sq.num_aggs;
            const uint32_t hist_mask = sq.hist_mask, fdirect_mask = sq.fdirect_mask;
            uint32_t *err = sq.err;
            for (int f = 0; f < nfac; f++) {
                if (pre || !((hist_mask >> f) & 1u) || ((fdirect_mask >> f) & 1u)) continue;
                const uint32_t nord = sq.sh.fa_nord[f];
                const uint32_t *off = sq.sh.fa_off[f];
                const uint32_t *ord = sq.sh.fa_ord[f];
                uint32_t *gh = sq.hist[f];
                uint32_t *tl = sq.touched[f], *tn = sq.ntouched[f];
                const bool use_lds = nord <= lds_bins;
                if (use_lds) {
                    for (uint32_t i = threadIdx.x; i < nord; i += MG_BLOCK) hist[i] = 0;
                    __syncthreads();
                }
                const uint32_t *ords = have_ords ? sq.fac_ord[f] : nullptr;
                for (uint32_t x = x0; x < x1; x++) {
                    const uint32_t vb = __builtin_amdgcn_readfirstlane(s_items[x].y);
                    if (vb >= nvb) continue;
                    for (uint32_t i = vb * MG_BLOCK + threadIdx.x; i < M; i += stride) {
                        // the rank kernel left the hit's ordinal behind (one value, the common case) ...
                        const uint32_t code = have_ords ? ords[i] : MG_FAC_LOOKUP;
                        if (code == MG_FAC_NONE) continue;
                        if (code != MG_FAC_LOOKUP) {
                            if (!mg_ck_always(err, 9, code, nord)) continue;              // permanent guard: see k_rank_tile, site 4
                            if (use_lds) atomicAdd(&hist[code], 1u);
                            else wave_append(atomicAdd(&gh[code], 1u) == 0, code, tl, tn);
                            continue;
                        }
                        // ... else the document's ordinals for field f come from its record when they fit there, else
                        // from the per-field CSR
                        const uint32_t d = hits[i];
                        const uint32_t *r = rec + (size_t)d * S4;
                        const uint32_t *vals; uint32_t cnt;
                        if ((r[3] >> 8) & 1u) {
                            const uint32_t *fp = r + 4 + 2 * nnum_in;
                            for (int g = 0; g < f; g++) fp += 1 + fp[0];
                            cnt = fp[0]; vals = fp + 1;
                        } else { const uint32_t a0 = off[d]; cnt = off[d + 1] - a0; vals = ord + a0; }
                        for (uint32_t j = 0; j < cnt; j++) {
                            const uint32_t o = vals[j];
                            if (use_lds) atomicAdd(&hist[o], 1u);
                            else wave_append(atomicAdd(&gh[o], 1u) == 0, o, tl, tn);          // first touch: remember the bin
                        }
                    }
                }
                if (use_lds) {
                    __syncthreads();
                    for (uint32_t i0 = 0; i0 < nord; i0 += MG_BLOCK) {
                        const uint32_t i = i0 + threadIdx.x;
                        const uint32_t c = i < nord ? hist[i] : 0;
                        wave_append(c && atomicAdd(&gh[i], c) == 0, i, tl, tn);
                    }
                    __syncthreads();
                }
            }

            // Aggregations: DP_AGGS of them per pass over the slice -- one load of the hit, the values of all of them in
            // flight together, one barrier pair -- where each used to make a pass of its own (the slices of the bench
            // stream are one or two hits per lane: the passes were dependent-load latencies in a row).  Per aggregation
            // the order of additions is what it was: the lane's hits in order, the fixed lane tree, the waves in order.
            // (fast && !pre: aggregated before the sample replaced the hit list)
            for (uint32_t x = x0; x < x1 && naggs > 0 && !(fast && !pre); x++) {
                const uint32_t vb = __builtin_amdgcn_readfirstlane(s_items[x].y);
                if (vb >= nvb) continue;
                for (int a0 = 0; a0 < naggs; a0 += DP_AGGS) {
                    double sum[DP_AGGS], mn[DP_AGGS], mx[DP_AGGS];
                    unsigned long long cnt[DP_AGGS];
                    const double *col[DP_AGGS]; const uint64_t *grp[DP_AGGS];
                    uint32_t af[DP_AGGS]; bool live[DP_AGGS], in_rec[DP_AGGS];
#pragma unroll
                    for (int u = 0; u < DP_AGGS; u++) {
                        const int a = a0 + u;
                        live[u] = a < naggs && sq.agg_type[a] != MGPU_A_CARDINALITY;
                        af[u] = live[u] ? (uint32_t)sq.agg_field[a] : 0u;
                        col[u] = live[u] ? sq.sh.num_col[af[u]] : nullptr;
                        grp[u] = (live[u] && fast) ? sq.sh.grp_has[af[u]] : nullptr;  // fast mode reads idnum2dbl: a missing group = no value
                        in_rec[u] = af[u] < nnum_in;
                        sum[u] = 0.0; mn[u] = 1.7976931348623157e308; mx[u] = -1.7976931348623157e308; cnt[u] = 0;
                    }
                    for (uint32_t i = vb * MG_BLOCK + threadIdx.x; i < M; i += stride) {
                        const uint32_t d = hits[i];
                        double v[DP_AGGS];
#pragma unroll
                        for (int u = 0; u < DP_AGGS; u++)
                            v[u] = !live[u] ? -1.7976931348623157e308 : in_rec[u] ? *(const double *)(rec + (size_t)d * S4 + 4 + 2 * af[u]) : col[u][d];
#pragma unroll
                        for (int u = 0; u < DP_AGGS; u++) {
                            if (!live[u]) continue;
                            if (grp[u] && !((grp[u][(d / 1000u) >> 6] >> ((d / 1000u) & 63u)) & 1ull)) continue;
                            if (v[u] != -1.7976931348623157e308) {       // get_num_value's "no value" marker, metric-aggs.c:106,149
                                sum[u] += v[u]; mn[u] = v[u] < mn[u] ? v[u] : mn[u]; mx[u] = v[u] > mx[u] ? v[u] : mx[u]; cnt[u]++;
                            }
                        }
                    }
                    // fixed-shape tree: lanes, then waves, so the f64 sum is reproducible run to run
#pragma unroll
                    for (int u = 0; u < DP_AGGS; u++) {
                        if (!live[u]) continue;                       // workgroup-uniform
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) {
                            const double s2 = __shfl_down(sum[u], o, 64), n2 = __shfl_down(mn[u], o, 64), x2 = __shfl_down(mx[u], o, 64);
                            const unsigned long long c2 = __shfl_down(cnt[u], o, 64);
                            sum[u] += s2; mn[u] = n2 < mn[u] ? n2 : mn[u]; mx[u] = x2 > mx[u] ? x2 : mx[u]; cnt[u] += c2;
                        }
                        if ((threadIdx.x & 63) == 0) { AggPart p = {sum[u], mn[u], mx[u], cnt[u]}; red[u][threadIdx.x >> 6] = p; }
                    }
                    __syncthreads();
                    if (threadIdx.x < DP_AGGS) {
                        const int a = a0 + (int)threadIdx.x;
                        if (a < naggs && sq.agg_type[a] != MGPU_A_CARDINALITY) {
                            AggPart p = red[threadIdx.x][0];
                            for (int w2 = 1; w2 < MG_BLOCK / 64; w2++) {
                                const AggPart q = red[threadIdx.x][w2];
                                p.sum += q.sum; p.mn = q.mn < p.mn ? q.mn : p.mn; p.mx = q.mx > p.mx ? q.mx : p.mx; p.cnt += q.cnt;
                            }
                            sq.agg_part[(size_t)vb * naggs + a] = p;
                        }
                    }
                    __syncthreads();
                }
            }
            x0 = x1;
        }
    }
}

// combines the per-slice partials in slice order; one wave per squery
__global__ __launch_bounds__(64) void k_aggfinal(const SqDev *sqs, int pre) {
    const SqDev &sq = sqs[blockIdx.x];
    const int a = threadIdx.x;
    if (a >= sq.num_aggs || sq.agg_type[a] == MGPU_A_CARDINALITY) return;
    const bool fast = sq.fast_eligible && *sq.fast_flag;
    if (fast != (pre != 0)) return;                  // a fast-ranked squery's partials belong to the pre-pass, the others' to the normal one
    const uint32_t M = *sq.num_hits;
    const uint32_t tiles = (M + MG_BLOCK - 1) / MG_BLOCK;
    const uint32_t nact = tiles < sq.nproc_blocks ? tiles : sq.nproc_blocks;
    AggPart p = {0.0, 1.7976931348623157e308, -1.7976931348623157e308, 0};
    // (eight partials loaded ahead, combined in the same order: the walk is one lane's chain of dependent loads otherwise)
    for (uint32_t b0 = 0; b0 < nact; b0 += 8) {
        AggPart q[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) if (b0 + u < nact) q[u] = sq.agg_part[(size_t)(b0 + u) * sq.num_aggs + a];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++)
            if (b0 + u < nact) { p.sum += q[u].sum; p.mn = q[u].mn < p.mn ? q[u].mn : p.mn; p.mx = q[u].mx > p.mx ? q[u].mx : p.mx; p.cnt += q[u].cnt; }
    }
    sq.agg_out[a] = p;
}

// touched facet bins -> selection keys: smaller key = larger count, ties by facet id (squery.c:411-437).
// Key i belongs to the i-th touched bin; the bits that differ between keys are recorded so the
// radix select skips the constant ones.
struct FkJob { const uint32_t *hist; const uint32_t *ord2id; const uint32_t *touched; const uint32_t *ntouched;
               u64 *key_hi, *key_lo; u64 *andor; };
__global__ __launch_bounds__(MG_BLOCK) void k_fkeys(const FkJob *jobs) {
    const FkJob j = jobs[blockIdx.y];
    const uint32_t n = *j.ntouched;
    const int lane = threadIdx.x & 63;
    u64 a = 0, b = 0, na = 0, nb = 0;
    for (uint32_t i = blockIdx.x * MG_BLOCK + threadIdx.x; i < n; i += gridDim.x * MG_BLOCK) {
        const uint32_t bin = j.touched[i];
        // (the count is taken with an exchange that leaves zero behind: self-cleaned pool, mgpu_ctx)
        const u64 hi = (u64)(0xFFFFFFFFu - atomicExch((uint32_t *)j.hist + bin, 0u)), lo = j.ord2id[bin];
        j.key_hi[i] = hi; j.key_lo[i] = lo;
        a |= hi; b |= lo; na |= ~hi; nb |= ~lo;
    }
    a = wave_or64(a); b = wave_or64(b); na = wave_or64(na); nb = wave_or64(nb);
    if (lane == 0 && (a | b | na | nb)) {
        atomicOr((u64 *)&j.andor[0], na); atomicOr((u64 *)&j.andor[1], nb);
        atomicOr((u64 *)&j.andor[2], a);  atomicOr((u64 *)&j.andor[3], b);
    }
}

// distinct facet ids among the hits (cardinality agg, metric-aggs.c:23-45,47-55): the touched bins
struct CardJob { const uint32_t *ord2id; const uint32_t *touched; const uint32_t *ntouched; uint32_t *out_ids; uint32_t *out_n;
                 uint32_t *clean_hist; };   // histogram to put back to zero (fields no facet job consumes), or NULL
__global__ __launch_bounds__(MG_BLOCK) void k_card(const CardJob *jobs) {
    const CardJob j = jobs[blockIdx.y];
    const uint32_t n = *j.ntouched;
    if (blockIdx.x == 0 && threadIdx.x == 0) *j.out_n = n;
    for (uint32_t i = blockIdx.x * MG_BLOCK + threadIdx.x; i < n; i += gridDim.x * MG_BLOCK)
    {
        const uint32_t bin = j.touched[i];
        j.out_ids[i] = j.ord2id[bin];
        if (j.clean_hist) j.clean_hist[bin] = 0;
    }
}

// ------------------------------------------------------------------ top-k selection
// Replaces rank_partial_sort + rank_sort (sort.c:134-214) and ks_partialsort over facets
// (squery.c:434): MSB-first radix select over only the key bits that differ between keys
// (plus the index, so ties resolve to the lowest docid / facet id), 11 bits per pass, with the
// candidate list compacted on every pass; a single workgroup then bitonic-sorts what is left.
__device__ __forceinline__ uint32_t key_bit(u64 hi, u64 lo, uint32_t idx, uint32_t b) {
    if (b < 64) return (uint32_t)(hi >> (63 - b)) & 1u;
    if (b < 128) return (uint32_t)(lo >> (127 - b)) & 1u;
    return (idx >> (159 - b)) & 1u;
}
__device__ __forceinline__ bool mask_bit(u64 mhi, u64 mlo, uint32_t midx, uint32_t b) {
    return key_bit(mhi, mlo, midx, b) != 0;
}
// positions of the next MG_SEL_DIGIT varying bits at or after `from`, written as (word << 8 | shift)
// with word 0 = hi, 1 = lo, 2 = idx; returns how many, *next = scan position after.
// Called by one thread; the table lives in LDS so that no kernel here needs private arrays.
__device__ __forceinline__ int digit_bits(u64 mhi, u64 mlo, uint32_t midx, uint32_t from, uint32_t *pos, uint32_t *next) {
    int n = 0;
    uint32_t b = from;                       // bit b of the 160-bit (hi, lo, idx) string, MSB first
    while (n < MG_SEL_DIGIT && b < 160) {
        if (b < 128) {
            const uint32_t base = b < 64 ? 0u : 64u, o = b - base;
            const u64 m = (b < 64 ? mhi : mlo) & (o == 0 ? ~0ull : ((1ull << (64 - o)) - 1ull));
            if (!m) { b = base + 64; continue; }
            b = base + (uint32_t)__clzll((long long)m);
            pos[n++] = base == 0 ? (63 - b) : ((1u << 8) | (127 - b));
            b++;
        } else {
            const uint32_t o = b - 128;
            const uint32_t m = midx & (o == 0 ? ~0u : ((1u << (32 - o)) - 1u));
            if (!m) { b = 160; break; }
            b = 128 + (uint32_t)__clz((int)m);
            pos[n++] = (2u << 8) | (159 - b);
            b++;
        }
    }
    *next = b;
    return n;
}
// the digit's bit positions held in registers (loaded once per pass from the LDS table)
struct DigitPos { uint32_t e[MG_SEL_DIGIT]; int n; };
__device__ __forceinline__ DigitPos load_digit_pos(const uint32_t *pos, int n) {
    DigitPos d; d.n = n;
#pragma unroll
    for (int j = 0; j < MG_SEL_DIGIT; j++) d.e[j] = __builtin_amdgcn_readfirstlane(pos[j]);   // uniform: keep it in SGPRs
    return d;
}
__device__ __forceinline__ uint32_t digit_of(u64 hi, u64 lo, uint32_t idx, const DigitPos &p) {
    uint32_t d = 0;
#pragma unroll
    for (int j = 0; j < MG_SEL_DIGIT; j++) {
        if (j < p.n) {
            const uint32_t e = p.e[j], w = e >> 8, sh = e & 0xFF;
            const u64 v = w == 0 ? hi : w == 1 ? lo : (u64)idx;
            d = (d << 1) | (uint32_t)((v >> sh) & 1u);
        }
    }
    return d;
}
__device__ __forceinline__ void job_masks(const SelJob &job, uint32_t M, u64 &mhi, u64 &mlo, uint32_t &midx) {
    if (job.mask) { mhi = job.mask[2] & job.mask[0]; mlo = job.mask[3] & job.mask[1]; }
    else { mhi = ~0ull; mlo = ~0ull; }
    midx = (M > 1) ? (0xFFFFFFFFu >> __clz(M - 1)) : 0u;
}

// Histogram update that survives skew: rank keys share most of their leading bits, so whole waves
// tend to land in a handful of bins and plain LDS atomics would serialise on them.  Up to SEL_PEEL
// times, the first pending lane's bin is counted for all lanes holding it (one atomic); whatever
// is left after that goes one atomic per lane.
#define SEL_PEEL 4
__device__ __forceinline__ void hist_add(uint32_t *hist, bool alive, uint32_t digit) {
    u64 rem = __ballot(alive);
#pragma unroll
    for (int r = 0; r < SEL_PEEL; r++) {
        if (!rem) break;                                                    // wave-uniform
        const int leader = __ffsll((long long)rem) - 1;
        const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)digit, leader);
        const u64 same = __ballot(alive && digit == dl);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[dl], (uint32_t)__popcll(same));
        alive = alive && digit != dl;
        rem &= ~same;
    }
    if (alive) atomicAdd(&hist[digit], 1u);
}
// wave-aggregated slot allocation: position of this lane's element in a list counted by *counter
__device__ __forceinline__ uint32_t wave_slot(bool pred, uint32_t *counter) {
    const u64 m = __ballot(pred);
    if (!m) return 0;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, __ffsll((long long)m) - 1);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
}

#define SEL_ITEMS 8
#define SEL_CHUNK (MG_BLOCK * SEL_ITEMS)     // candidates per workgroup iteration
#define SEL_SOLO_CHUNKS 4                    // a lone workgroup takes lists up to 8192 and runs every remaining pass in one launch
#define SEL_SMALL_CAP MG_SEL_SMALL            // final sort sizes: small / big workgroup variant
#define SEL_BIG_CAP (2 * MG_SEL_FINAL)
// extra ctl words used by the passes
enum { SEL_NIN = SEL_NCAND, SEL_DSTAR = SEL_NEXT, SEL_BPREV = 8, SEL_NOUT = 9, SEL_BUF = 10, SEL_NPREV = 11, SEL_NFIN = 12, SEL_NB = 13 };

__device__ __forceinline__ uint32_t sel_nblocks(uint32_t nin, uint32_t gdim) {
    const uint32_t nch = (nin + SEL_CHUNK - 1) / SEL_CHUNK;
    if (nch <= SEL_SOLO_CHUNKS) return 1;
    return nch < gdim ? nch : gdim;
}

struct SelOut { u64 *out_hi, *out_lo; uint32_t *out_pl; const uint32_t *payload; };

// bitonic sort of the s_cnt[0] keys gathered in LDS by (hi, lo, idx), then emit the first k
template <int CAP, int NT = MG_BLOCK>
__device__ __forceinline__ void sel_sort_emit(const SelJob &job, const SelOut &out, u64 *s_hi, u64 *s_lo, uint32_t *s_ix, uint32_t *s_cnt) {
    __syncthreads();
    const uint32_t n = s_cnt[0] < (uint32_t)CAP ? s_cnt[0] : (uint32_t)CAP;
    uint32_t np2 = 1; while (np2 < n) np2 <<= 1;
    for (uint32_t i = n + threadIdx.x; i < np2; i += NT) { s_hi[i] = ~0ull; s_lo[i] = ~0ull; s_ix[i] = 0xFFFFFFFFu; }
    __syncthreads();
    for (uint32_t k = 2; k <= np2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < np2; i += NT) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;
                    const u64 ah = s_hi[i], al = s_lo[i], bh = s_hi[l], bl = s_lo[l];
                    const uint32_t ai = s_ix[i], bi = s_ix[l];
                    const bool gt = (ah > bh) || (ah == bh && (al > bl || (al == bl && ai > bi)));
                    if (gt == up) { s_hi[i] = bh; s_lo[i] = bl; s_ix[i] = bi; s_hi[l] = ah; s_lo[l] = al; s_ix[l] = ai; }
                }
            }
            __syncthreads();
        }
    }
    const uint32_t kk = job.k < n ? job.k : n;
    for (uint32_t i = threadIdx.x; i < kk; i += NT) {
        if (job.skip_ones && s_hi[i] >= 0xFFFFFFFFull) continue;        // facet entries with a zero count (see SqDev::fdirect_mask)
        // derived facet keys carry the bin ordinal (same order as the facet id: ordinals are ranks of
        // the sorted ids); the id itself is looked up for the k winners only
        out.out_hi[i] = s_hi[i]; out.out_lo[i] = job.gen_touched ? (u64)job.gen_ord2id[s_lo[i]] : s_lo[i];
        out.out_pl[i] = out.payload ? out.payload[s_ix[i]] : s_ix[i];
        atomicAdd(&s_cnt[1], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) *job.out_n = s_cnt[1];
}

// The same result without a sort, for sets of at most CAP <= 512 keys: every key counts the keys before it (the
// (hi, lo, idx) order is total), which IS its place in the output; no barrier in the counting.  Two keys per lane at
// most for 256 lanes, so the loop over the set runs once or twice.
template <int CAP, int NT = MG_BLOCK>
__device__ __forceinline__ void sel_rank_emit(const SelJob &job, const SelOut &out, u64 *s_hi, u64 *s_lo, uint32_t *s_ix, uint32_t *s_cnt) {
    __syncthreads();
    const uint32_t n = s_cnt[0] < (uint32_t)CAP ? s_cnt[0] : (uint32_t)CAP;
    for (uint32_t c = threadIdx.x; c < n; c += NT) {
        const u64 h = s_hi[c], l = s_lo[c]; const uint32_t ix = s_ix[c];
        if (job.skip_ones && h >= 0xFFFFFFFFull) continue;             // facet entries with a zero count (see SqDev::fdirect_mask): they sort last
        uint32_t r = 0;
#pragma unroll 8
        for (uint32_t j = 0; j < n; j++) r += key_lt(s_hi[j], s_lo[j], s_ix[j], h, l, ix) ? 1u : 0u;
        if (r < job.k) {
            // derived facet keys carry the bin ordinal (same order as the facet id: ordinals are ranks of
            // the sorted ids); the id itself is looked up for the k winners only
            out.out_hi[r] = h; out.out_lo[r] = job.gen_touched ? (u64)job.gen_ord2id[l] : l;
            out.out_pl[r] = out.payload ? out.payload[ix] : ix;
            atomicAdd(&s_cnt[1], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *job.out_n = s_cnt[1];
}

// Final step of a selection, run by one workgroup: gather the winners the passes found and the
// last candidate list with the last decision applied, bitonic-sort them by (hi, lo, idx) in LDS and
// emit the first k.  s_cnt: two LDS words.
// REGS: the whole candidate list (one chunk) is still in the caller's registers: no second read of the keys
template <int CAP, bool REGS = false>
__device__ __forceinline__ void sel_final(const SelJob &job, const SelOut &out, uint32_t pass, uint32_t nin, uint32_t nwin,
                                          uint32_t dstar, const uint32_t *rin, const uint32_t *pprev_lds, int nprev,
                                          u64 *s_hi, u64 *s_lo, uint32_t *s_ix, uint32_t *s_cnt,
                                          const uint32_t *ridx = nullptr, const u64 *rhi = nullptr, const u64 *rlo = nullptr) {
    if (threadIdx.x == 0) { s_cnt[0] = 0; s_cnt[1] = 0; }
    __syncthreads();
    const DigitPos pprev = load_digit_pos(pprev_lds, nprev);
    for (uint32_t i = threadIdx.x; i < nwin; i += MG_BLOCK) {
        const uint32_t idx = job.out_idx[i];
        const uint32_t p = atomicAdd(&s_cnt[0], 1u);
        if (p < CAP) { s_hi[p] = job.key_hi[idx]; s_lo[p] = job.key_lo[idx]; s_ix[p] = idx; }
    }
    if (REGS) {
#pragma unroll
        for (int u = 0; u < SEL_ITEMS; u++) {
            bool take = ridx[u] != 0xFFFFFFFFu;
            if (take && pass > 0) take = digit_of(rhi[u], rlo[u], ridx[u], pprev) <= dstar;
            if (take) {
                const uint32_t p = atomicAdd(&s_cnt[0], 1u);
                if (p < CAP) { s_hi[p] = rhi[u]; s_lo[p] = rlo[u]; s_ix[p] = ridx[u]; }
            }
        }
    } else
    for (uint32_t i0 = threadIdx.x; i0 < nin; i0 += 4 * MG_BLOCK) {
        uint32_t idxs[4]; u64 his[4], los[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t i = i0 + u * MG_BLOCK; idxs[u] = (i < nin) ? (rin ? rin[i] : i) : 0xFFFFFFFFu; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool v = idxs[u] != 0xFFFFFFFFu;
            his[u] = v ? job.key_hi[idxs[u]] : 0; los[u] = v ? job.key_lo[idxs[u]] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            bool take = idxs[u] != 0xFFFFFFFFu;
            if (take && pass > 0) take = digit_of(his[u], los[u], idxs[u], pprev) <= dstar;
            if (take) {
                const uint32_t p = atomicAdd(&s_cnt[0], 1u);
                if (p < CAP) { s_hi[p] = his[u]; s_lo[p] = los[u]; s_ix[p] = idxs[u]; }
            }
        }
    }
    sel_sort_emit<CAP>(job, out, s_hi, s_lo, s_ix, s_cnt);
}

// One pass = filter the candidate list by the previous digit decision (winners below it go to the
// output, the digit's own bucket stays a candidate) and histogram the survivors on the next digit;
// the last workgroup to finish picks the bucket holding the k-th key.  Lists that fit one
// workgroup (the common case: a shard's hits for one query) never come back to the host: that
// workgroup keeps iterating passes in LDS until the answer fits the final sort, and when that
// sort is small it does it on the spot (SEL_DONE = 2): no further launch touches the job.
// Device-side work lists of the select stage (all in the batch arena):
//   items[p] / nitems[p]: {job, workgroup index} pairs that the p-th wide launch runs (filled by the
//   launch before it: a job asks for exactly the workgroups its current list needs);
//   fin / nfin: jobs whose final set needs the stand-alone sort kernels.
#define SEL_MAX_WG 64
#define SEL_WIDE_LAUNCHES 3
struct SelLists { uint2 *items[SEL_WIDE_LAUNCHES + 1]; uint32_t *nitems; uint32_t *fin; uint32_t *nfin; uint32_t cap_items;
                  uint32_t *mid; uint32_t *nmid; int lanemin; };
enum { SELMODE_HOST = 0, SELMODE_ITEMS = 2 };

// The digit's bit positions folded into runs of adjacent bits inside one 32-bit word of the key
// (word 0..3 = hi.hi, hi.lo, lo.hi, lo.lo, 4 = idx), MSB first: (word << 16 | low bit << 8 | length).
// A run costs two instructions per key (bit-field extract, shift-or) where the per-bit form costs six.
__device__ __forceinline__ int digit_runs(const uint32_t *pos, int n, uint32_t *runs) {
    int nr = 0; uint32_t cw = 0, clow = 0, clen = 0;
    for (int j = 0; j < n; j++) {
        const uint32_t e = pos[j], w = e >> 8, sh = e & 0xFF;
        const uint32_t w32 = (w == 2) ? 4u : (w * 2 + (sh >= 32 ? 0u : 1u)), bit = sh & 31u;
        if (clen && w32 == cw && bit + 1 == clow) { clow = bit; clen++; }
        else { if (clen) runs[nr++] = cw << 16 | clow << 8 | clen; cw = w32; clow = bit; clen = 1; }
    }
    if (clen) runs[nr++] = cw << 16 | clow << 8 | clen;
    return nr;
}

// A fresh list that one workgroup can hold (M <= SEL_SOLO_CHUNKS * SEL_CHUNK) with a small k: the
// whole selection in one go and without touching memory between passes.  Nothing is compacted:
// every pass looks at all M keys again (from registers when they fit SEL_SOLO_REG chunks, else a
// coalesced re-read that L2 serves) and keeps those whose decided prefix bits equal the prefix
// chosen so far (pfx under mask msk, both over the varying bits consumed); the keys at or below the
// last prefix are the final set.  Same decisions as the compacting passes, no lists, no fences.
#define SEL_SOLO_REG 1
#define SEL_MID_KEYS MG_SEL_MID_KEYS
#define SEL_MID_BLOCK 1024
struct Pfx { u64 hi, lo; uint32_t ix; };
template <int NT>
__device__ __forceinline__ void select_solo(const SelJob &job, const SelOut &out, uint32_t M, u64 mhi, u64 mlo, uint32_t midx,
                                            uint32_t *hist, uint32_t *s_scan, uint32_t *s_pcur, uint32_t *s_runs,
                                            uint32_t *s_misc /* >= 6 */, u64 *s_fhi, u64 *s_flo, uint32_t *s_fix, uint32_t *s_fcnt,
                                            u64 *s_red /* >= 4 * NT / 64 */, bool pregen = false) {
    DBG_T(ts0);
    constexpr int DS = (NT == MG_BLOCK) ? 4 : 17;        // phase-clock slots: small / big workgroup variant
    constexpr int R = SEL_SOLO_REG * SEL_ITEMS;
    constexpr uint32_t SEL_SOLO_GROUP = (uint32_t)R * NT;           // keys held in registers at a time
    const uint32_t ngrp = (M + SEL_SOLO_GROUP - 1) / SEL_SOLO_GROUP;
    const bool reg = ngrp == 1;
    u64 rhi[R], rlo[R];
#define SOLO_LOAD(g) do { _Pragma("unroll") for (int u = 0; u < R; u++) { \
        const uint32_t i = (g) * SEL_SOLO_GROUP + u * NT + threadIdx.x; \
        rhi[u] = (i < M) ? job.key_hi[i] : 0; rlo[u] = (i < M) ? job.key_lo[i] : 0; } } while (0)
    // facet keys derived here from the touched bins: (0xFFFFFFFF - count, bin ordinal), loads batched R deep.
    // The count is taken with an exchange that leaves zero behind: the histogram pool cleans itself (mgpu_ctx).
#define SOLO_GEN(g) do { uint32_t bins[R]; _Pragma("unroll") for (int u = 0; u < R; u++) { \
        const uint32_t i = (g) * SEL_SOLO_GROUP + u * NT + threadIdx.x; bins[u] = (i < M) ? job.gen_touched[i] : 0xFFFFFFFFu; } \
        _Pragma("unroll") for (int u = 0; u < R; u++) { const bool v = bins[u] < MG_FAC_LOOKUP; /* a bin, not "no value" */ \
        rhi[u] = v ? (u64)(0xFFFFFFFFu - atomicExch((uint32_t *)job.gen_hist + bins[u], 0u)) : 0xFFFFFFFFull; rlo[u] = (u64)bins[u]; } } while (0)
    if (job.gen_touched && !pregen) {
        // the varying bits of these keys are not known yet: OR / AND them over the list first.  A list
        // beyond the registers is written to key_hi/key_lo on the way, and read back from there after.
        u64 a = 0, b = 0, na = 0, nb = 0;
        for (uint32_t g = 0; g < ngrp; g++) {
            SOLO_GEN(g);
#pragma unroll
            for (int u = 0; u < R; u++) {
                const uint32_t i = g * SEL_SOLO_GROUP + u * NT + threadIdx.x;
                if (i < M) {
                    a |= rhi[u]; b |= rlo[u]; na |= ~rhi[u]; nb |= ~rlo[u];
                    if (!reg) { ((u64 *)job.key_hi)[i] = rhi[u]; ((u64 *)job.key_lo)[i] = rlo[u]; }
                }
            }
        }
        a = wave_or64(a); b = wave_or64(b); na = wave_or64(na); nb = wave_or64(nb);
        if ((threadIdx.x & 63) == 0) { u64 *r = s_red + 4 * (threadIdx.x >> 6); r[0] = a; r[1] = b; r[2] = na; r[3] = nb; }
        if (!reg) __threadfence();
        __syncthreads();
        a = b = na = nb = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) { a |= s_red[4 * w]; b |= s_red[4 * w + 1]; na |= s_red[4 * w + 2]; nb |= s_red[4 * w + 3]; }
        mhi = a & na; mlo = b & nb;
    } else if (reg) SOLO_LOAD(0);
#undef SOLO_GEN
    Pfx pfx = {0, 0, 0}, msk = {0, 0, 0};
    uint32_t bitpos = 0, nbelow = 0;                 // keys strictly below the prefix: all of them are in the top k
    constexpr uint32_t BPT = MG_SEL_BINS / NT;
    DBG_T(ts1); DBG_ADD(DS, ts0, ts1);
    for (;;) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < MG_SEL_BINS; i += NT) hist[i] = 0;
        if (threadIdx.x == 0) {
            uint32_t nb2;
            const int nc = digit_bits(mhi, mlo, midx, bitpos, s_pcur, &nb2);
            s_misc[0] = (uint32_t)nc; s_misc[5] = (uint32_t)digit_runs(s_pcur, nc, s_runs);
            s_misc[1] = nb2; s_misc[2] = 0xFFFFFFFFu; s_misc[3] = 0; s_misc[4] = 0;
        }
        __syncthreads();
        const int ncur = (int)__builtin_amdgcn_readfirstlane(s_misc[0]);
        bitpos = __builtin_amdgcn_readfirstlane(s_misc[1]);
        const int nruns = (int)__builtin_amdgcn_readfirstlane(s_misc[5]);
        const DigitPos dcur = load_digit_pos(s_pcur, ncur);
        DBG_T(tc0);
        for (uint32_t g = 0; g < ngrp; g++) {
            if (!reg) SOLO_LOAD(g);
            uint32_t dg[R];
#pragma unroll
            for (int u = 0; u < R; u++) dg[u] = 0;
            for (int r = 0; r < nruns; r++) {                       // uniform: the word choice is a scalar branch
                const uint32_t e = __builtin_amdgcn_readfirstlane(s_runs[r]);
                const uint32_t w32 = e >> 16, low = (e >> 8) & 0xFFu, len = e & 0xFFu;
#define SOLO_RUN(expr) _Pragma("unroll") for (int u = 0; u < R; u++) dg[u] = (dg[u] << len) | __builtin_amdgcn_ubfe((uint32_t)(expr), low, len)
                if (w32 == 0) { SOLO_RUN(rhi[u] >> 32); }
                else if (w32 == 1) { SOLO_RUN(rhi[u]); }
                else if (w32 == 2) { SOLO_RUN(rlo[u] >> 32); }
                else if (w32 == 3) { SOLO_RUN(rlo[u]); }
                else { SOLO_RUN(g * SEL_SOLO_GROUP + u * NT + threadIdx.x); }
#undef SOLO_RUN
            }
#pragma unroll
            for (int u = 0; u < R; u++) {
                const uint32_t i = g * SEL_SOLO_GROUP + u * NT + threadIdx.x;
                hist_add(hist, i < M && (rhi[u] & msk.hi) == pfx.hi && (rlo[u] & msk.lo) == pfx.lo && (i & msk.ix) == pfx.ix, dg[u]);
            }
        }
        __syncthreads();
        DBG_T(tc1); DBG_ADD(DS + 1, tc0, tc1);
        // the bin where the running count of the surviving keys crosses what is still wanted
        const uint32_t krem = job.k - nbelow;
        uint32_t loc[BPT], lsum = 0;
#pragma unroll
        for (uint32_t j = 0; j < BPT; j++) { loc[j] = hist[threadIdx.x * BPT + j]; lsum += loc[j]; }
        uint32_t tot;
        const uint32_t excl = block_excl_scan<NT>(lsum, &tot, s_scan);
        if (excl < krem && excl + lsum >= krem) {
            uint32_t acc = excl;
#pragma unroll
            for (uint32_t j = 0; j < BPT; j++) {
                if (acc < krem && acc + loc[j] >= krem) { s_misc[2] = threadIdx.x * BPT + j; s_misc[3] = acc + loc[j]; s_misc[4] = acc; }
                acc += loc[j];
            }
        }
        __syncthreads();
        uint32_t d = __builtin_amdgcn_readfirstlane(s_misc[2]), cum = __builtin_amdgcn_readfirstlane(s_misc[3]);
        uint32_t below = __builtin_amdgcn_readfirstlane(s_misc[4]);
        if (d >= (1u << ncur)) { d = (1u << ncur) - 1; cum = tot; below = tot; }       // cannot happen while k < M; stay in range
        // extend the prefix by this digit
#pragma unroll
        for (int j = 0; j < MG_SEL_DIGIT; j++) {
            if (j < ncur) {
                const uint32_t e = dcur.e[j], w = e >> 8, sh = e & 0xFF, bit = (d >> (ncur - 1 - j)) & 1u;
                if (w == 0) { msk.hi |= 1ull << sh; pfx.hi |= (u64)bit << sh; }
                else if (w == 1) { msk.lo |= 1ull << sh; pfx.lo |= (u64)bit << sh; }
                else { msk.ix |= 1u << sh; pfx.ix |= bit << sh; }
            }
        }
        DBG_T(tc2); DBG_ADD(DS + 2, tc1, tc2); if (NT != MG_BLOCK) DBG_CNT(22);
        if (nbelow + cum <= SEL_SMALL_CAP || ncur == 0) break;
        nbelow += below;
    }
    // final set: every key at or below the prefix
    DBG_T(tf0);
    if (threadIdx.x == 0) { s_fcnt[0] = 0; s_fcnt[1] = 0; }
    __syncthreads();
    for (uint32_t g = 0; g < ngrp; g++) {
        if (!reg) SOLO_LOAD(g);
#pragma unroll
        for (int u = 0; u < R; u++) {
            const uint32_t i = g * SEL_SOLO_GROUP + u * NT + threadIdx.x;
            const u64 a = rhi[u] & msk.hi, b = rlo[u] & msk.lo; const uint32_t c = i & msk.ix;
            const bool take = i < M && (a < pfx.hi || (a == pfx.hi && (b < pfx.lo || (b == pfx.lo && c <= pfx.ix))));
            const uint32_t p = wave_slot(take, &s_fcnt[0]);
            if (take && p < SEL_SMALL_CAP) { s_fhi[p] = rhi[u]; s_flo[p] = rlo[u]; s_fix[p] = i; }
        }
    }
#undef SOLO_LOAD
    if (threadIdx.x == 0) job.ctl[SEL_DONE] = 2u;            // later launches and host rounds leave the job alone
    sel_rank_emit<SEL_SMALL_CAP, NT>(job, out, s_fhi, s_flo, s_fix, s_fcnt);
    DBG_T(tf1); DBG_ADD(DS + 3, tf0, tf1); DBG_CNT(DS + 4); if (NT == MG_BLOCK) { DBG_MAX(9, ts0, tf1, M); DBG_MAX(10, tf0, tf1, M); } else { DBG_MAX(11, ts0, tf1, M); DBG_MAX(12, tf0, tf1, M); }
}

// Small k out of a list one workgroup holds (a shard's hits for one query, top 10): no radix passes at all.
// Lane l owns keys l, l + 256, ... (R = ceil(M / 256) of them).  The k-th smallest of the 256 per-lane MINIMA, T, is an
// upper bound of the k-th smallest key (k lanes hold a key <= T), and a key <= T can only sit in one of the k lanes whose
// minimum is <= T: at most k * R keys survive the cut "key <= T", and the top k are among them.  One pass for the
// minima, a counting rank over 256 minima in LDS (no barriers inside), one pass for the survivors, then the small sort.
// The order is (hi, lo, index) as everywhere in this file, so ties resolve to the lowest docid / facet id.
// Requires k <= NT < M and k * R <= SEL_SMALL_CAP (checked by the caller).
// Rank of this lane's key (h, l, i) among the NT lanes' keys when it is one of the k smallest, else some number >= k.
// Keys must differ between lanes.  Two levels: every wave ranks its 64 keys with lane broadcasts (no LDS), the min(k, 64)
// smallest of each wave -- the only ones that can be among the k smallest overall -- meet in LDS and are ranked against
// each other.  b_hi / b_lo / b_ix: (NT / 64) * min(k, 64) entries each.  Contains three barriers.
template <int NT>
__device__ __forceinline__ uint32_t lane_rank_below_k(u64 h, u64 l, uint32_t i, uint32_t k, u64 *b_hi, u64 *b_lo, uint32_t *b_ix) {
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t lr = 0;
#pragma unroll
    for (int j = 0; j < 64; j++) {
        const u64 oh = __shfl(h, j, 64), ol = __shfl(l, j, 64);
        const uint32_t oi = __shfl(i, j, 64);
        lr += key_lt(oh, ol, oi, h, l, i) ? 1u : 0u;
    }
    const uint32_t kk = k < 64 ? k : 64;
    if (lr < kk) { b_hi[wave * kk + lr] = h; b_lo[wave * kk + lr] = l; b_ix[wave * kk + lr] = i; }
    __syncthreads();
    // Candidate c is ranked by thread c: the (NT / 64) * kk candidates fill whole waves, where their owners are kk lanes
    // of EVERY wave (all NT / 64 waves would walk the list); the rank goes back to the owner through b_ix.
    const uint32_t nc = (NT / 64) * kk;
    uint32_t r = 0;
    if (threadIdx.x < nc) {
        const u64 ch = b_hi[threadIdx.x], cl = b_lo[threadIdx.x]; const uint32_t ci = b_ix[threadIdx.x];
#pragma unroll 8
        for (uint32_t x = 0; x < nc; x++) r += key_lt(b_hi[x], b_lo[x], b_ix[x], ch, cl, ci) ? 1u : 0u;
    }
    __syncthreads();
    if (threadIdx.x < nc) b_ix[threadIdx.x] = r;
    __syncthreads();
    return lr < kk ? b_ix[wave * kk + lr] : k;
}

template <int NT>
__device__ __forceinline__ void select_lanemin(const SelJob &job, const SelOut &out, uint32_t M,
                                               u64 *b_hi /* >= (NT / 64) * 64 */, u64 *b_lo, uint32_t *b_ix, uint32_t *s_misc /* >= 6 */,
                                               u64 *s_fhi, u64 *s_flo, uint32_t *s_fix, uint32_t *s_fcnt, bool pregen = false) {
    constexpr int R = SEL_ITEMS;                                   // keys per lane held in registers at a time
    constexpr uint32_t GROUP = (uint32_t)R * NT;
    const uint32_t ngrp = (M + GROUP - 1) / GROUP;
    const bool reg = ngrp == 1;
    const bool gen = job.gen_touched != nullptr && !pregen;        // pregen: derived facet keys already sit in key_hi / key_lo
    u64 rhi[R], rlo[R];
    u64 mh = ~0ull, ml = ~0ull; uint32_t mi = 0xFFFFFFFFu;         // this lane's least key
    for (uint32_t g = 0; g < ngrp; g++) {
        if (gen) {
            // facet keys derived from the touched bins: (0xFFFFFFFF - count, bin ordinal); the count is taken with an
            // exchange that leaves zero behind (self-cleaned pool); a list beyond the registers is parked in key_hi/lo
            uint32_t bins[R];
#pragma unroll
            for (int u = 0; u < R; u++) { const uint32_t i = g * GROUP + u * NT + threadIdx.x; bins[u] = (i < M) ? job.gen_touched[i] : 0xFFFFFFFFu; }
#pragma unroll
            for (int u = 0; u < R; u++) {
                const bool v = bins[u] < MG_FAC_LOOKUP;
                rhi[u] = v ? (u64)(0xFFFFFFFFu - atomicExch((uint32_t *)job.gen_hist + bins[u], 0u)) : 0xFFFFFFFFull;
                rlo[u] = (u64)bins[u];
                const uint32_t i = g * GROUP + u * NT + threadIdx.x;
                if (!reg && i < M) { ((u64 *)job.key_hi)[i] = rhi[u]; ((u64 *)job.key_lo)[i] = rlo[u]; }
            }
        } else {
#pragma unroll
            for (int u = 0; u < R; u++) {
                const uint32_t i = g * GROUP + u * NT + threadIdx.x;
                rhi[u] = (i < M) ? job.key_hi[i] : ~0ull; rlo[u] = (i < M) ? job.key_lo[i] : ~0ull;
            }
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
            const uint32_t i = g * GROUP + u * NT + threadIdx.x;
            if (i < M && key_lt(rhi[u], rlo[u], i, mh, ml, mi)) { mh = rhi[u]; ml = rlo[u]; mi = i; }
        }
    }
    if (threadIdx.x == 0) { s_fcnt[0] = 0; s_fcnt[1] = 0; }
    if (gen && !reg) __threadfence();
    // rank of this lane's minimum among the NT minima (all distinct: M > NT, and the index is part of the key)
    const uint32_t rank = lane_rank_below_k<NT>(mh, ml, mi, job.k, b_hi, b_lo, b_ix);
    if (rank == job.k - 1) { s_fhi[SEL_SMALL_CAP - 1] = mh; s_flo[SEL_SMALL_CAP - 1] = ml; s_misc[0] = mi; }   // T; exactly one lane
    __syncthreads();
    const u64 th = s_fhi[SEL_SMALL_CAP - 1], tlo = s_flo[SEL_SMALL_CAP - 1]; const uint32_t ti = s_misc[0];
    __syncthreads();                                                // (the slot is an ordinary output slot again from here on)
    // survivors: keys <= T, only in lanes whose minimum is <= T
    const bool mine = rank < job.k;
    for (uint32_t g = 0; g < ngrp; g++) {
        if (!reg) {
#pragma unroll
            for (int u = 0; u < R; u++) {
                const uint32_t i = g * GROUP + u * NT + threadIdx.x;
                const bool v = mine && i < M;
                rhi[u] = v ? job.key_hi[i] : ~0ull; rlo[u] = v ? job.key_lo[i] : ~0ull;
            }
        }
#pragma unroll
        for (int u = 0; u < R; u++) {
            const uint32_t i = g * GROUP + u * NT + threadIdx.x;
            const bool take = mine && i < M && !key_lt(th, tlo, ti, rhi[u], rlo[u], i);
            const uint32_t p = wave_slot(take, &s_fcnt[0]);
            if (take && p < SEL_SMALL_CAP) { s_fhi[p] = rhi[u]; s_flo[p] = rlo[u]; s_fix[p] = i; }
        }
    }
    if (threadIdx.x == 0) job.ctl[SEL_DONE] = 2u;                   // later launches and host rounds leave the job alone
    sel_rank_emit<SEL_SMALL_CAP, NT>(job, out, s_fhi, s_flo, s_fix, s_fcnt);
}

// mode HOST : host-driven continuation, grid (gx, njobs): workgroup count from gridDim.x, vb = blockIdx.x
// mode ITEMS: wide launch `slot` over items[slot]; this workgroup is number vb of ctl[SEL_NB]
__device__ __forceinline__ void select_job(const SelJob &job, const SelOut &sel_out, int mode, uint32_t job_index,
                                           uint32_t vb, const SelLists &L, int slot) {
    uint32_t *ctl = job.ctl;
    const uint32_t M = *job.m_ptr;
    __shared__ u64 s_fhi[SEL_SMALL_CAP], s_flo[SEL_SMALL_CAP];
    __shared__ uint32_t s_fix[SEL_SMALL_CAP], s_fcnt[2];
    DBG_T(ts0);
    DbgScope dbg_scope(13); dbg_scope.tag = M;
    // until a final sort has run the host must see "not finished"
    if (vb == 0 && threadIdx.x == 0 &&
        __hip_atomic_load(&ctl[SEL_DONE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 2u) *job.out_n = 0xFFFFFFFFu;
    __shared__ uint32_t hist[MG_SEL_BINS];
    __shared__ uint32_t s_ctl[SEL_WORDS];
    __shared__ uint32_t s_scan[8];
    __shared__ uint32_t s_last, s_nout, s_nwin, s_dstar, s_cum;
    __shared__ uint32_t s_pprev[MG_SEL_DIGIT], s_pcur[MG_SEL_DIGIT], s_nprev, s_ncur, s_nextbit;
    u64 mhi, mlo; uint32_t midx;
    job_masks(job, M, mhi, mlo, midx);
    const int lane = threadIdx.x & 63;
    uint32_t vbid = vb;
    bool continuing = false;
    __shared__ uint32_t s_at;

    for (;;) {
        __syncthreads();
        if (threadIdx.x < SEL_WORDS) s_ctl[threadIdx.x] = __hip_atomic_load(&ctl[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ctl[SEL_DONE]) return;
        const uint32_t pass = s_ctl[SEL_PASS];
        const uint32_t nin = (pass == 0) ? M : s_ctl[SEL_NIN];
        if (pass == 0 && (M <= SEL_SMALL_CAP || job.k >= M)) {
            if (vb != 0) return;
            const bool small = M <= SEL_SMALL_CAP;
            if (threadIdx.x == 0) {
                ctl[SEL_NIN] = M; ctl[SEL_KREM] = job.k; ctl[SEL_NFIN] = M; ctl[SEL_DONE] = small ? 2u : 1u;
                if (!small && mode != SELMODE_HOST) L.fin[atomicAdd(L.nfin, 1u)] = job_index;
            }
            if (small) sel_final<SEL_SMALL_CAP>(job, sel_out, 0, M, 0, 0, nullptr, s_pcur, 0, s_fhi, s_flo, s_fix, s_fcnt);
            return;
        }
        const uint32_t nb = continuing ? 1u : mode == SELMODE_ITEMS ? s_ctl[SEL_NB] : sel_nblocks(nin, gridDim.x);
        if (!continuing && vbid >= nb) return;
        const bool solo = nb == 1;

        for (uint32_t i = threadIdx.x; i < MG_SEL_BINS; i += MG_BLOCK) hist[i] = 0;
        const uint32_t bitpos = (pass == 0) ? 0 : s_ctl[SEL_BITPOS];
        if (threadIdx.x == 0) {
            s_nout = 0; s_nwin = s_ctl[SEL_NWIN]; s_dstar = 0xFFFFFFFFu; s_cum = 0;
            uint32_t dummy, nb2;
            s_nprev = (pass > 0) ? (uint32_t)digit_bits(mhi, mlo, midx, s_ctl[SEL_BPREV], s_pprev, &dummy) : 0u;
            s_ncur = (uint32_t)digit_bits(mhi, mlo, midx, bitpos, s_pcur, &nb2);
            s_nextbit = nb2;
        }
        const uint32_t dstar = s_ctl[SEL_DSTAR];
        const uint32_t buf = s_ctl[SEL_BUF];
        const uint32_t *rin = (pass <= 1) ? nullptr : (buf ? job.cand[1] : job.cand[0]);
        uint32_t *rout = ((buf ^ (pass <= 1 ? 0u : 1u)) & 1u) ? job.cand[1] : job.cand[0];
        uint32_t *win = job.out_idx;        // winners accumulate in the output buffer, sorted at the end
        __syncthreads();
        const int nprev = (int)s_nprev, ncur = (int)s_ncur;
        const uint32_t next_bitpos = s_nextbit;
        const DigitPos dprev = load_digit_pos(s_pprev, nprev), dcur = load_digit_pos(s_pcur, ncur);
        DBG_T(ts1); DBG_ADD(4, ts0, ts1);

        const uint32_t nch = (nin + SEL_CHUNK - 1) / SEL_CHUNK;
        uint32_t idxs[SEL_ITEMS]; u64 his[SEL_ITEMS], los[SEL_ITEMS];       // outlive the loop: see the fused final below
        for (uint32_t ch = vbid; ch < nch; ch += nb) {
            // all loads of the chunk first (8 candidates per lane in flight), then the decisions
#pragma unroll
            for (int it = 0; it < SEL_ITEMS; it++) {
                const uint32_t i = ch * SEL_CHUNK + it * MG_BLOCK + threadIdx.x;
                idxs[it] = (i < nin) ? (rin ? rin[i] : i) : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int it = 0; it < SEL_ITEMS; it++) {
                const bool v = idxs[it] != 0xFFFFFFFFu;
                his[it] = v ? job.key_hi[idxs[it]] : 0; los[it] = v ? job.key_lo[idxs[it]] : 0;
            }
#pragma unroll
            for (int it = 0; it < SEL_ITEMS; it++) {
                const uint32_t idx = idxs[it]; const u64 hi = his[it], lo = los[it];
                bool keep = idx != 0xFFFFFFFFu, winner = false;
                if (keep && pass > 0) {
                    const uint32_t dp = digit_of(hi, lo, idx, dprev);
                    if (dp < dstar) { winner = true; keep = false; }
                    else if (dp > dstar) keep = false;
                }
                if (pass > 0) {
                    // wave-aggregated appends
                    const u64 mk = __ballot(keep), mw = __ballot(winner);
                    uint32_t bk = 0, bw = 0;
                    if (lane == 0) {
                        if (solo) {
                            if (mk) bk = atomicAdd(&s_nout, (uint32_t)__popcll(mk));
                            if (mw) bw = atomicAdd(&s_nwin, (uint32_t)__popcll(mw));
                        } else {
                            if (mk) bk = atomicAdd(&ctl[SEL_NOUT], (uint32_t)__popcll(mk));
                            if (mw) bw = atomicAdd(&ctl[SEL_NWIN], (uint32_t)__popcll(mw));
                        }
                    }
                    bk = __shfl(bk, 0, 64); bw = __shfl(bw, 0, 64);
                    const u64 below = (1ull << lane) - 1;
                    if (keep) rout[bk + (uint32_t)__popcll(mk & below)] = idx;
                    if (winner) win[bw + (uint32_t)__popcll(mw & below)] = idx;
                }
                hist_add(hist, keep, digit_of(hi, lo, idx, dcur));
            }
        }
        __syncthreads();
        DBG_T(ts2); DBG_ADD(5, ts1, ts2);
        uint32_t nwin_total, nkept;
        if (!solo) {
            for (uint32_t i = threadIdx.x; i < MG_SEL_BINS; i += MG_BLOCK) {
                const uint32_t c = hist[i];
                if (c) atomicAdd(&job.hist[i], c);
            }
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) s_last = (atomicAdd(&ctl[SEL_TICKET], 1u) == nb - 1) ? 1u : 0u;
            __syncthreads();
            if (!s_last) return;
            // last workgroup of this pass
            __threadfence();
            for (uint32_t i = threadIdx.x; i < MG_SEL_BINS; i += MG_BLOCK) {
                hist[i] = __hip_atomic_load(&job.hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                job.hist[i] = 0;
            }
            nwin_total = __hip_atomic_load(&ctl[SEL_NWIN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nkept = (pass == 0) ? M : __hip_atomic_load(&ctl[SEL_NOUT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        } else {
            nwin_total = s_nwin;
            nkept = (pass == 0) ? M : s_nout;
        }
        // winners appended so far are exactly the keys below every decided digit
        const uint32_t krem_now = job.k - nwin_total;
        // parallel search of the bin where the running count crosses krem_now: 8 bins per thread
        const uint32_t nbins = 1u << ncur;
        constexpr uint32_t BPT = MG_SEL_BINS / MG_BLOCK;
        uint32_t loc[BPT], lsum = 0;
#pragma unroll
        for (uint32_t j = 0; j < BPT; j++) { loc[j] = hist[threadIdx.x * BPT + j]; lsum += loc[j]; }
        uint32_t tot;
        const uint32_t excl = block_excl_scan(lsum, &tot, s_scan);
        if (excl < krem_now && excl + lsum >= krem_now) {
            uint32_t acc = excl;
#pragma unroll
            for (uint32_t j = 0; j < BPT; j++) {
                if (acc < krem_now && acc + loc[j] >= krem_now) { s_dstar = threadIdx.x * BPT + j; s_cum = acc + loc[j]; }
                acc += loc[j];
            }
        }
        __syncthreads();
        bool done = false, go_on = false, fused = false;
        uint32_t d_fin = 0, newbuf = buf;
        {
            uint32_t d = s_dstar, cum = s_cum;
            if (d >= nbins) { d = nbins - 1; cum = nkept; }
            // the final workgroup streams the kept list and sorts what passes the last decision
            // stop as soon as what is left fits the final sort: the small in-place one whenever k allows
            // it (a 512-element sort costs a fraction of the 4096-element one), else the big one
            const uint32_t fin_cap = (job.k <= SEL_SMALL_CAP / 2) ? SEL_SMALL_CAP : MG_SEL_FINAL;
            if (pass > 0 && nkept <= fin_cap && nwin_total + nkept <= 2 * fin_cap) done = true;
            else if (nkept <= SEL_SOLO_CHUNKS * SEL_CHUNK && nwin_total + cum <= fin_cap) done = true;
            go_on = !done && (mode == SELMODE_HOST ? sel_nblocks(nkept, gridDim.x) == 1
                                                   : nkept <= SEL_SOLO_CHUNKS * SEL_CHUNK);
            fused = done && nkept <= SEL_SOLO_CHUNKS * SEL_CHUNK && nwin_total + cum <= SEL_SMALL_CAP;
            d_fin = d;
            if (pass >= 1) newbuf = buf ^ (pass == 1 ? 0u : 1u);
            if (threadIdx.x == 0) {
                ctl[SEL_DSTAR] = d;
                ctl[SEL_BPREV] = bitpos;
                ctl[SEL_BITPOS] = next_bitpos;
                ctl[SEL_KREM] = krem_now;
                ctl[SEL_NIN] = nkept;
                ctl[SEL_NWIN] = nwin_total;
                ctl[SEL_NOUT] = 0;
                ctl[SEL_TICKET] = 0;
                ctl[SEL_NPREV] = (uint32_t)ncur;
                ctl[SEL_NFIN] = nwin_total + cum;
                ctl[SEL_BUF] = newbuf;
                ctl[SEL_PASS] = pass + 1;
                if (done) ctl[SEL_DONE] = fused ? 2u : 1u;
            }
        }
        DBG_T(ts3); DBG_ADD(6, ts2, ts3);
        if (fused) {
            // this workgroup is the only one left on the job: sort and emit right here
            __threadfence();
            __syncthreads();
            if (pass == 0 && nch == 1 && solo)       // the only chunk is still in registers
                sel_final<SEL_SMALL_CAP, true>(job, sel_out, 1, nkept, nwin_total, d_fin, nullptr, s_pcur, ncur,
                                               s_fhi, s_flo, s_fix, s_fcnt, idxs, his, los);
            else
                sel_final<SEL_SMALL_CAP>(job, sel_out, pass + 1, nkept, nwin_total, d_fin,
                                         (pass + 1 <= 1) ? nullptr : (newbuf ? job.cand[1] : job.cand[0]), s_pcur, ncur, s_fhi, s_flo, s_fix, s_fcnt);
            DBG_T(ts4); DBG_ADD(7, ts3, ts4); DBG_CNT(8); DBG_MAX(9, ts0, ts4, M); DBG_MAX(10, ts3, ts4, M);
            return;
        }
        if (done && !fused && mode != SELMODE_HOST && threadIdx.x == 0) L.fin[atomicAdd(L.nfin, 1u)] = job_index;
        if (!done && !go_on && mode == SELMODE_ITEMS && slot + 1 <= SEL_WIDE_LAUNCHES) {
            // still too long for one workgroup: order the workgroups of the next wide launch
            const uint32_t want = sel_nblocks(nkept, SEL_MAX_WG);
            if (threadIdx.x == 0) {
                const uint32_t at = atomicAdd(&L.nitems[slot + 1], want);
                s_at = (at + want <= L.cap_items) ? at : 0xFFFFFFFFu;
                ctl[SEL_NB] = want;
            }
            __syncthreads();
            if (s_at != 0xFFFFFFFFu && threadIdx.x < want) L.items[slot + 1][s_at + threadIdx.x] = make_uint2(job_index, threadIdx.x);
        }
        if (!go_on) return;
        // next pass in this launch: make this pass's list and control words visible to the whole workgroup
        __threadfence();
        continuing = true; vbid = 0;
    }
}

// ------------------------------------------------------------------ sparse facet field, few hits: count by hashing
// process_facet_result + sort_facets (docrank.c:371-389, squery.c:411-437) for a field with far more values than the
// squery has hits (100k buckets, a few thousand hits).  The hits' ordinals (fac_ord, left behind by the rank kernel)
// are counted in an LDS hash table -- no global histogram, no atomics outside the CU -- in ONE pass: the table has at
// least twice as many slots as the squery has hits, so linear probing always finds room.  The table's best k (count
// desc, ordinal asc = facet id asc) are cut out with the lane-minimum bound of select_lanemin (16 slots per lane: at
// most 16 k keys survive) and put in order by counting ranks.  Two sizes of the same kernel: 256 lanes / 4096 slots for
// up to MG_FHASH_SMALL_M hits, 1024 lanes / 16384 slots up to MG_FHASH_MAX_M; a job runs in the one its M selects.
#define FH_PER_LANE 16
template <int NT>
__global__ __launch_bounds__(NT) void k_facet_hash(const SelJob *jobs, const SelOut *outs, const uint32_t *list, const uint32_t *count, uint32_t cap) {
    constexpr uint32_t SLOTS = (uint32_t)NT * FH_PER_LANE;
    constexpr uint32_t CAND = MG_FHASH_MAX_K * FH_PER_LANE;
    __shared__ uint32_t t_key[SLOTS], t_cnt[SLOTS];
    __shared__ u64 s_bhi[(NT / 64) * MG_FHASH_MAX_K], s_blo[(NT / 64) * MG_FHASH_MAX_K];
    __shared__ uint32_t s_bix[(NT / 64) * MG_FHASH_MAX_K];
    __shared__ u64 s_cand[CAND];
    __shared__ uint32_t s_n[2];
    __shared__ u64 s_T;
    uint32_t n = *count; if (n > cap) n = cap;
    for (uint32_t w = blockIdx.x; w < n; w += gridDim.x) {
        const uint32_t ji = __builtin_amdgcn_readfirstlane(list[w]);
        const SelJob job = jobs[ji];
        const SelOut out = outs[ji];
        const uint32_t M = *job.m_ptr;
        const uint32_t k = job.k;
        const uint32_t *ords = job.gen_touched;
        for (uint32_t i = threadIdx.x; i < SLOTS; i += NT) { t_key[i] = 0xFFFFFFFFu; t_cnt[i] = 0; }
        if (threadIdx.x == 0) { s_n[0] = 0; s_T = 0xFFFFFFFFFFFFFFFEull; }
        __syncthreads();
        for (uint32_t i0 = 0; i0 < M; i0 += 4 * NT) {
            uint32_t o[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t i = i0 + u * NT + threadIdx.x; o[u] = (i < M) ? ords[i] : 0xFFFFFFFFu; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (o[u] >= MG_FAC_LOOKUP) continue;              // the document has no value in the field
                uint32_t sl = hash_wid(o[u] + 1u) & (SLOTS - 1);
                for (;; sl = (sl + 1) & (SLOTS - 1)) {            // ends: fewer distinct values than slots (M <= SLOTS / 2)
                    const uint32_t old = atomicCAS(&t_key[sl], 0xFFFFFFFFu, o[u]);
                    if (old == 0xFFFFFFFFu || old == o[u]) { atomicAdd(&t_cnt[sl], 1u); break; }
                }
            }
        }
        __syncthreads();
        // this lane's slots as keys ((~count) << 32 | ordinal; empty = ~0), and their minimum
        u64 key[FH_PER_LANE], mn = ~0ull;
#pragma unroll
        for (int u = 0; u < FH_PER_LANE; u++) {
            const uint32_t sl = u * NT + threadIdx.x;
            const uint32_t kk = t_key[sl];
            key[u] = kk == 0xFFFFFFFFu ? ~0ull : (((u64)(0xFFFFFFFFu - t_cnt[sl])) << 32) | kk;
            mn = key[u] < mn ? key[u] : mn;
        }
        // rank of the minimum among the lanes' minima (the lane number makes empty lanes distinct)
        const uint32_t rank = lane_rank_below_k<NT>(mn, 0ull, threadIdx.x, k, s_bhi, s_blo, s_bix);
        if (rank == k - 1 && mn != ~0ull) s_T = mn;               // the k-th smallest minimum bounds the k-th smallest key
        __syncthreads();
        const u64 T = s_T;
#pragma unroll
        for (int u = 0; u < FH_PER_LANE; u++) {
            const bool take = rank < k && key[u] <= T;
            const uint32_t at = wave_slot(take, &s_n[0]);
            if (take && at < CAND) s_cand[at] = key[u];
        }
        __syncthreads();
        // counting rank: a candidate with r smaller ones is the r-th of the result (every key is distinct: one per ordinal)
        const uint32_t nc = s_n[0] < CAND ? s_n[0] : CAND;
        for (uint32_t c = threadIdx.x; c < nc; c += NT) {
            const u64 mine = s_cand[c];
            uint32_t r = 0;
#pragma unroll 8
            for (uint32_t j = 0; j < nc; j++) r += s_cand[j] < mine ? 1u : 0u;
            if (r < k) {
                out.out_hi[r] = mine >> 32;
                out.out_lo[r] = (u64)job.gen_ord2id[(uint32_t)mine];
                out.out_pl[r] = 0;
            }
        }
        if (threadIdx.x == 0) { *job.out_n = nc < k ? nc : k; job.ctl[SEL_DONE] = 2u; }
        __syncthreads();                                          // LDS is reused by the next job
    }
}

// host-driven continuation rounds: grid (gx, njobs)
__global__ __launch_bounds__(MG_BLOCK) void k_select_pass(const SelJob *jobs, const SelOut *outs, int mode, SelLists L) {
    const SelJob job = jobs[blockIdx.y];
    select_job(job, outs[blockIdx.y], mode, blockIdx.y, blockIdx.x, L, 0);
}
// First launch of a batch's select stage: one workgroup per job, control words known to be zero (the
// arena was just cleared).  Short lists are sorted outright, lists one workgroup can hold are selected
// by select_solo, everything else asks for its workgroups in the first wide launch (a list that fits
// one workgroup but wants a big k goes there too, as a one-workgroup item).
__global__ __launch_bounds__(MG_BLOCK) void k_select_first(const SelJob *jobs, const SelOut *outs, SelLists L) {
    const uint32_t ji = blockIdx.x;
    const SelJob job = jobs[ji];
    const SelOut out = outs[ji];
    const uint32_t M = *job.m_ptr;
    __shared__ u64 s_fhi[SEL_SMALL_CAP], s_flo[SEL_SMALL_CAP];
    __shared__ uint32_t s_fix[SEL_SMALL_CAP], s_fcnt[2];
    __shared__ __align__(16) uint32_t hist[MG_SEL_BINS];
    __shared__ uint32_t s_scan[8], s_pcur[MG_SEL_DIGIT], s_runs[MG_SEL_DIGIT], s_misc[8];
    __shared__ u64 s_red[4 * MG_BLOCK / 64];
    DbgScope dbg_scope(13); dbg_scope.tag = M;
    if (job.fhash && M <= MG_FHASH_MAX_M) return;                // k_facet_hash's job
    if (M <= SEL_SMALL_CAP) {
        // short list: straight into the sort
        if (threadIdx.x == 0) { job.ctl[SEL_NIN] = M; job.ctl[SEL_KREM] = job.k; job.ctl[SEL_NFIN] = M; job.ctl[SEL_DONE] = 2u;
                                s_fcnt[0] = M; s_fcnt[1] = 0; }
        uint32_t bins[SEL_SMALL_CAP / MG_BLOCK];
#pragma unroll
        for (int u = 0; u < SEL_SMALL_CAP / MG_BLOCK; u++) {
            const uint32_t i = u * MG_BLOCK + threadIdx.x;
            bins[u] = (job.gen_touched && i < M) ? job.gen_touched[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < SEL_SMALL_CAP / MG_BLOCK; u++) {
            const uint32_t i = u * MG_BLOCK + threadIdx.x;
            if (i < M) {
                // (the count is taken with an exchange that leaves zero behind: self-cleaned pool, mgpu_ctx)
                s_fhi[i] = !job.gen_touched ? job.key_hi[i]
                         : bins[u] >= MG_FAC_LOOKUP ? 0xFFFFFFFFull : (u64)(0xFFFFFFFFu - atomicExch((uint32_t *)job.gen_hist + bins[u], 0u));
                s_flo[i] = job.gen_touched ? (u64)bins[u] : job.key_lo[i];
                s_fix[i] = i;
            }
        }
        sel_rank_emit<SEL_SMALL_CAP>(job, out, s_fhi, s_flo, s_fix, s_fcnt);
        return;
    }
    if (threadIdx.x == 0) *job.out_n = 0xFFFFFFFFu;          // until a final sort has run the host must see "not finished"
    if (job.k >= M) {                                         // everything is wanted: the big stand-alone sort
        if (threadIdx.x == 0) {
            job.ctl[SEL_NIN] = M; job.ctl[SEL_KREM] = job.k; job.ctl[SEL_NFIN] = M; job.ctl[SEL_DONE] = 1u;
            L.fin[atomicAdd(L.nfin, 1u)] = ji;
        }
        return;
    }
    if (M <= SEL_SOLO_CHUNKS * SEL_CHUNK && job.k <= MG_BLOCK && job.k * ((M + MG_BLOCK - 1) / MG_BLOCK) <= SEL_SMALL_CAP && L.lanemin) {
        // (hist doubles as the minima's LDS: 3 x 256 words of key + 256 of index fit its 2048 words)
        select_lanemin<MG_BLOCK>(job, out, M, (u64 *)hist, (u64 *)hist + MG_BLOCK, hist + 4 * MG_BLOCK, s_misc, s_fhi, s_flo, s_fix, s_fcnt);
        return;
    }
    if (M <= SEL_SOLO_CHUNKS * SEL_CHUNK && job.k <= MG_SEL_SOLO_K) {
        u64 mhi, mlo; uint32_t midx;
        job_masks(job, M, mhi, mlo, midx);
        select_solo<MG_BLOCK>(job, out, M, mhi, mlo, midx, hist, s_scan, s_pcur, s_runs, s_misc, s_fhi, s_flo, s_fix, s_fcnt, s_red);
        return;
    }
    // same selection by a 1024-thread workgroup of k_select_mid (a list whose keys are derived has no other place to go:
    // the wide launches read keys, they do not make them)
    if ((M <= SEL_MID_KEYS || job.gen_touched) && job.k <= MG_SEL_SOLO_K) {
        if (threadIdx.x == 0) L.mid[atomicAdd(L.nmid, 1u)] = ji;
        return;
    }
    const uint32_t want = sel_nblocks(M, SEL_MAX_WG);
    if (threadIdx.x == 0) {
        const uint32_t at = atomicAdd(&L.nitems[0], want);
        s_misc[0] = (at + want <= L.cap_items) ? at : 0xFFFFFFFFu;       // no room: the host rounds pick the job up
        job.ctl[SEL_NB] = want;
    }
    __syncthreads();
    if (s_misc[0] != 0xFFFFFFFFu && threadIdx.x < want) L.items[0][s_misc[0] + threadIdx.x] = make_uint2(ji, threadIdx.x);
}
// Derived facet keys of the mid-size lists: (0xFFFFFFFF - count, bin ordinal) written to key_hi / key_lo by
// SEL_FKM_WG workgroups per list (the one big workgroup of k_select_mid would spend most of its time in
// these gathers), varying bits accumulated into the job's mask words as k_fkeys does.
#define SEL_FKM_WG 16
__global__ __launch_bounds__(MG_BLOCK) void k_fkeys_mid(const SelJob *jobs, SelLists L) {
    const uint32_t n = *L.nmid;
    const int lane = threadIdx.x & 63;
    for (uint32_t w = blockIdx.y; w < n; w += gridDim.y) {
        const SelJob job = jobs[L.mid[w]];
        if (!job.gen_touched) continue;
        const uint32_t M = *job.m_ptr;
        u64 a = 0, b = 0, na = 0, nb = 0;
        for (uint32_t i = blockIdx.x * MG_BLOCK + threadIdx.x; i < M; i += gridDim.x * MG_BLOCK) {
            const uint32_t bin = job.gen_touched[i];
            const u64 hi = bin >= MG_FAC_LOOKUP ? 0xFFFFFFFFull : (u64)(0xFFFFFFFFu - atomicExch((uint32_t *)job.gen_hist + bin, 0u)), lo = bin;
            ((u64 *)job.key_hi)[i] = hi; ((u64 *)job.key_lo)[i] = lo;
            a |= hi; b |= lo; na |= ~hi; nb |= ~lo;
        }
        a = wave_or64(a); b = wave_or64(b); na = wave_or64(na); nb = wave_or64(nb);
        if (lane == 0 && (a | b | na | nb)) {
            u64 *mk = (u64 *)job.mask;
            atomicOr(&mk[0], na); atomicOr(&mk[1], nb); atomicOr(&mk[2], a); atomicOr(&mk[3], b);
        }
    }
}
// The lists k_select_first found too long for 256 threads but short enough for one big workgroup:
// select_solo again, four times the lanes.  A small grid strides over the list.
__global__ __launch_bounds__(SEL_MID_BLOCK) void k_select_mid(const SelJob *jobs, const SelOut *outs, SelLists L) {
    __shared__ u64 s_fhi[SEL_SMALL_CAP], s_flo[SEL_SMALL_CAP];
    __shared__ uint32_t s_fix[SEL_SMALL_CAP], s_fcnt[2];
    __shared__ uint32_t hist[MG_SEL_BINS];
    __shared__ uint32_t s_scan[SEL_MID_BLOCK / 64], s_pcur[MG_SEL_DIGIT], s_runs[MG_SEL_DIGIT], s_misc[8];
    __shared__ u64 s_red[4 * SEL_MID_BLOCK / 64];
    __shared__ u64 s_bhi[SEL_MID_BLOCK], s_blo[SEL_MID_BLOCK];          // lane_rank_below_k: (1024 / 64) * 64 entries
    __shared__ uint32_t s_bix[SEL_MID_BLOCK];
    const uint32_t n = *L.nmid;
    for (uint32_t w = blockIdx.x; w < n; w += gridDim.x) {
        const uint32_t ji = L.mid[w];
        const SelJob job = jobs[ji];
        const uint32_t M = *job.m_ptr;
        // (derived facet keys of these long lists were written out by k_fkeys_mid, many workgroups per list)
        if (L.lanemin && job.k * ((M + SEL_MID_BLOCK - 1) / SEL_MID_BLOCK) <= SEL_SMALL_CAP) {
            select_lanemin<SEL_MID_BLOCK>(job, outs[ji], M, s_bhi, s_blo, s_bix, s_misc, s_fhi, s_flo, s_fix, s_fcnt, true);
        } else {
            u64 mhi, mlo; uint32_t midx;
            job_masks(job, M, mhi, mlo, midx);
            select_solo<SEL_MID_BLOCK>(job, outs[ji], M, mhi, mlo, midx, hist, s_scan, s_pcur, s_runs, s_misc, s_fhi, s_flo, s_fix, s_fcnt, s_red, true);
        }
        __syncthreads();
    }
}
// wide launch `slot`: a 1-D grid strides over items[slot]; enqueued behind the first launch without a
// host round trip, and nearly free when the list is empty
__global__ __launch_bounds__(MG_BLOCK) void k_select_items(const SelJob *jobs, const SelOut *outs, SelLists L, int slot) {
    uint32_t n = L.nitems[slot];
    if (n > L.cap_items) n = L.cap_items;
    for (uint32_t w = blockIdx.x; w < n; w += gridDim.x) {
        const uint2 item = L.items[slot][w];
        const uint32_t ji = __builtin_amdgcn_readfirstlane(item.x), vb = __builtin_amdgcn_readfirstlane(item.y);
        const SelJob job = jobs[ji];
        select_job(job, outs[ji], SELMODE_ITEMS, ji, vb, L, slot);
        __syncthreads();
    }
}

// final stage for the jobs the pass kernel did not finish itself (large final sets, or lists that
// needed several workgroups): one workgroup per job.  Two sizes of the same kernel.
template <int CAP, int LO>
__global__ __launch_bounds__(MG_BLOCK) void k_select_final(const SelJob *jobs, const SelOut *outs,
                                                           const uint32_t *fin_list, const uint32_t *fin_count) {
    // with a list: one launch of min(njobs, 256) workgroups serves the listed jobs only
    uint32_t ji = blockIdx.x;
    if (fin_list) { if (blockIdx.x >= *fin_count) return; ji = fin_list[blockIdx.x]; }
    const SelJob job = jobs[ji];
    const uint32_t *ctl = job.ctl;
    const uint32_t M = *job.m_ptr;
    __shared__ u64 s_hi[CAP];
    __shared__ u64 s_lo[CAP];
    __shared__ uint32_t s_ix[CAP];
    __shared__ uint32_t s_cnt[2];
    __shared__ uint32_t s_pprev[MG_SEL_DIGIT], s_nprev;
    const uint32_t done = ctl[SEL_DONE];
    if (done != 1u) return;                                      // 0: the host launches more passes; 2: already emitted
    const uint32_t nfin = ctl[SEL_NFIN];
    if (nfin > (uint32_t)CAP || nfin < (uint32_t)LO) return;     // the other variant's job
    const uint32_t pass = ctl[SEL_PASS];
    u64 mhi, mlo; uint32_t midx;
    job_masks(job, M, mhi, mlo, midx);
    if (threadIdx.x == 0) {
        uint32_t dummy;
        s_nprev = (pass > 0) ? (uint32_t)digit_bits(mhi, mlo, midx, ctl[SEL_BPREV], s_pprev, &dummy) : 0u;
    }
    __syncthreads();
    sel_final<CAP>(job, outs[ji], pass, ctl[SEL_NIN], ctl[SEL_NWIN], ctl[SEL_DSTAR],
                   (pass <= 1) ? nullptr : (ctl[SEL_BUF] ? job.cand[1] : job.cand[0]), s_pprev, (int)s_nprev, s_hi, s_lo, s_ix, s_cnt);
}

// ------------------------------------------------------------------ primitives for the parity tests
// bmap_and / bmap_andnot / oper_or + cardinality on two materialised rows
__global__ __launch_bounds__(MG_BLOCK) void k_row_op(const u64 *a, const u64 *b, u64 *out, uint32_t W, int op,
                                                     unsigned long long *card) {
    uint32_t c = 0;
    for (uint32_t w = blockIdx.x * MG_BLOCK + threadIdx.x; w < W; w += gridDim.x * MG_BLOCK) {
        const u64 x = a[w], y = b[w];
        const u64 v = (op == 0) ? (x & y) : (op == 1) ? (x | y) : (x & ~y);
        out[w] = v;
        c += (uint32_t)__popcll(v);
    }
    c = wave_sum32(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(card, (unsigned long long)c);
}

// ------------------------------------------------------------------ launch wrappers (called from mgpu_api.hip)
__global__ void k_nop(int *p) { if (p) *p = 0; }
__global__ void k_clock_cal(unsigned long long *out) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long w1 = w0;
    while (w1 - w0 < 1000) w1 = wall_clock64();          // bounded: the wall clock always advances
    out[0] = w1 - w0; out[1] = clock64() - c0;
}
// ------------------------------------------------------------------ f-1: term lookup
// dtrie_lookup_term's typo walk (dtrie.c:710-791) as a scan in two steps.
// k_term_filter: blockIdx.y = dictionary, blockIdx.x = 1024-word chunk of it; every lane holds the 4-byte signatures
// (mg_lev.h: character classes + length) of FOUR words and tests them against every term of the call that reads this
// dictionary (term signatures staged in LDS, read as broadcasts): ~10 instructions per (term, word) pair instead of the
// walk's several hundred.  The pairs that pass -- a necessary condition, about 1 in 100 -- go to a candidate list with
// one atomic per wave.
// k_term_eval: one lane per candidate scores the word with the walk's own row arithmetic (mg_lev_word) and appends the
// matches {lookup, wid, distance} to the hit list.  Both counters keep counting past their caps, so the host learns the
// sizes it needs.
#define TF_TERMS 256
#define TF_WBUF 512                                  // candidates a wave collects in LDS before it takes a range of the global list
#define TF_WORDS 4                                   // words per lane: the ~25 instructions of loop, LDS read and broadcast per (wave, term)
                                                     // are most of this kernel; four words share them
__global__ __launch_bounds__(MG_BLOCK) void k_term_filter(const TermJob *jobs, const TermDict *dicts, uint2 *cand, uint32_t *count, uint32_t cap) {
    __shared__ uint2 s_t[TF_TERMS];                  // {tsig, maxdist | prefix << 8}
    // One counter for the whole list, bumped by every wave that found something, serialises the chip on one address
    // (measured: 6 ms per call, as slow as walking every word).  Each wave therefore collects in a buffer of its own --
    // filled by its own lanes at ballot ranks, read back by the same wave: no barrier -- and takes a range of the list only when the
    // buffer fills up and once at the end.
    __shared__ uint2 s_buf[MG_BLOCK / 64][TF_WBUF];
    const TermDict *dp = &dicts[blockIdx.y];
    const uint32_t nwords = dp->nwords, job_lo = dp->job_lo, job_hi = dp->job_hi;
    if (blockIdx.x * (MG_BLOCK * TF_WORDS) >= nwords) return;
    uint32_t w[TF_WORDS], wsig[TF_WORDS];
#pragma unroll
    for (int u = 0; u < TF_WORDS; u++) {
        w[u] = (blockIdx.x * TF_WORDS + u) * MG_BLOCK + threadIdx.x;
        wsig[u] = w[u] < nwords ? dp->sig[w[u]] : 0u;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint2 *buf = s_buf[wv];
    uint32_t n = 0;                                  // entries in buf: the same value in every lane of the wave (ballots)
    auto flush = [&]() {
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(count, n);
        at = __shfl(at, 0, 64);
        for (uint32_t x = lane; x < n; x += 64) if (at + x < cap) cand[at + x] = buf[x];
    };
    for (uint32_t j0 = job_lo; j0 < job_hi; j0 += TF_TERMS) {
        const uint32_t nt = job_hi - j0 < TF_TERMS ? job_hi - j0 : TF_TERMS;
        __syncthreads();
        if (threadIdx.x < nt) {
            const TermJob *jp = &jobs[j0 + threadIdx.x];
            s_t[threadIdx.x] = make_uint2(jp->tsig, (uint32_t)jp->maxdist | ((uint32_t)jp->prefix << 8));
        }
        __syncthreads();
        for (uint32_t i = 0; i < nt; i++) {
            const uint2 t = s_t[i];
#pragma unroll
            for (int u = 0; u < TF_WORDS; u++) {
                const bool pass = (w[u] < nwords) & mg_lev_sig_pass(t.x, (int)(t.y & 0xFF), (int)(t.y >> 8), wsig[u]);
                const u64 m = __ballot(pass);
                if (!m) continue;
                const uint32_t add = (uint32_t)__popcll(m);
                if (n + add > TF_WBUF) { flush(); n = 0; }
                if (pass) buf[n + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = make_uint2(j0 + i, w[u]);
                n += add;
            }
        }
    }
    if (n) flush();
}
__global__ __launch_bounds__(MG_BLOCK) void k_term_eval(const TermJob *jobs, const uint32_t *term_chars, const uint2 *cand, const uint32_t *cand_count,
                                                        uint32_t cand_cap, uint32_t *hits, uint32_t *count, uint32_t cap) {
    uint32_t n = *cand_count;
    if (n > cand_cap) n = cand_cap;                 // the host runs the call again with a longer list
    const int lane = threadIdx.x & 63;
    for (uint32_t base = blockIdx.x * MG_BLOCK; base < n; base += gridDim.x * MG_BLOCK) {
        const uint32_t x = base + threadIdx.x;
        uint32_t dist = MG_LEV_NOMATCH, wid = 0, lookup = 0;
        if (x < n) {
            const uint2 cd = cand[x];
            const TermJob *jp = &jobs[cd.x];
            const uint32_t *off = jp->off;
            const uint32_t o0 = off[cd.y], o1 = off[cd.y + 1];
            // the walk ends by itself a few characters past the term's length; the bounds only keep the loop finite and
            // inside the array on a damaged offset table (mgpu_dict_create checks the table)
            int wlen = o1 > o0 ? (int)(o1 - o0) : 0;
            if (wlen > 4096) wlen = 4096;
            int tlen = jp->tlen;
            if (tlen > MGPU_MAX_TERM_LEN) tlen = MGPU_MAX_TERM_LEN;
            if (wlen > 0) dist = mg_lev_word(term_chars + jp->term_off, tlen, jp->maxdist, jp->prefix, jp->chars + o0, wlen);
            wid = jp->wid[cd.y]; lookup = jp->lookup;
        }
        const bool hit = dist != MG_LEV_NOMATCH;
        const u64 m = __ballot(hit);
        if (!m) continue;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t at = 0;
        if (lane == leader) at = atomicAdd(count, (uint32_t)__popcll(m));
        at = __shfl(at, leader, 64);
        if (hit) {
            const uint32_t idx = at + (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (idx < cap) { hits[3 * (size_t)idx] = lookup; hits[3 * (size_t)idx + 1] = wid; hits[3 * (size_t)idx + 2] = dist; }
        }
    }
}

extern "C" {
// diagnostic: ticks of wall_clock64 and of the shader clock over the same interval, plus the host's view in ns
int mgk_debug_clock_cal(unsigned long long *out3) {
    unsigned long long *d = nullptr;
    if (hipMalloc(&d, 16) != hipSuccess) return -1;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipFree(d); return -1; }
    hipLaunchKernelGGL(k_clock_cal, dim3(1), dim3(1), 0, 0, d);      // warm
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_clock_cal, dim3(1), dim3(1), 0, 0, d);
    (void)hipEventRecord(e1, 0);
    float ms = 0;
    const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess &&
                    hipMemcpy(out3, d, 16, hipMemcpyDeviceToHost) == hipSuccess;
    out3[2] = (unsigned long long)(ms * 1e6);
    (void)hipFree(d); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ok ? 0 : -1;
}
int mgk_debug_skip(int mask) { return hipMemcpyToSymbol(HIP_SYMBOL(mg_dbg_skip), &mask, sizeof(int)) == hipSuccess ? 0 : -1; }
// phase clocks: enable, and read + reset the 16 accumulators (100 MHz wall_clock64 ticks)
int mgk_debug_phases(int on, unsigned long long *out16) {
    unsigned long long z[32] = {0};
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(mg_dbg_acc), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(mg_dbg_acc), z, sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(mg_dbg_on), &on, sizeof(int)) != hipSuccess) return -1;
    return 0;
}
// counts[0] = hits, counts[1] = candidates (both zeroed by the caller)
void mgk_term_scan(hipStream_t st, const TermJob *jobs, const TermDict *dicts, uint32_t ndicts, uint32_t max_words, const uint32_t *term_chars,
                   uint2 *cand, uint32_t cand_cap, uint32_t *hits, uint32_t *counts, uint32_t cap) {
    if (!ndicts || !max_words) return;
    hipLaunchKernelGGL(k_term_filter, dim3((max_words + MG_BLOCK * TF_WORDS - 1) / (MG_BLOCK * TF_WORDS), ndicts), dim3(MG_BLOCK), 0, st, jobs, dicts, cand, counts + 1, cand_cap);
    hipLaunchKernelGGL(k_term_eval, dim3(1024), dim3(MG_BLOCK), 0, st, jobs, term_chars, (const uint2 *)cand, (const uint32_t *)(counts + 1), cand_cap, hits, counts, cap);
}
void mgk_range_rows(hipStream_t st, const RangeRow *rows, uint32_t nrows, uint32_t max_w) {
    if (!nrows) return;
    // A wave takes 64 words per step, a workgroup 256.  As many workgroups per group of rows as cover the row in the fewest
    // steps the budget of ~16k workgroups allows, and no more than that many steps need: every wave then runs the same
    // number of steps (64 workgroups over a row of 305 wave-steps left half the waves a second step to wait for)
    const uint32_t ngroups = (nrows + MG_RANGE_GROUP - 1) / MG_RANGE_GROUP;
    const uint32_t nw64 = (max_w + 63) / 64 ? (max_w + 63) / 64 : 1;
    const uint32_t gx_cap = 16384 / (ngroups < 16384 ? ngroups : 16384) ? 16384 / (ngroups < 16384 ? ngroups : 16384) : 1;
    const uint32_t steps = (nw64 + 4 * gx_cap - 1) / (4 * gx_cap);
    const uint32_t gx = (nw64 + 4 * steps - 1) / (4 * steps);
    for (uint32_t at = 0; at < ngroups; at += 65535) {
        const uint32_t n = ngroups - at < 65535 ? ngroups - at : 65535;
        hipLaunchKernelGGL(k_range_rows, dim3(gx, n), dim3(MG_BLOCK), 0, st, rows + (size_t)at * MG_RANGE_GROUP, nrows - at * MG_RANGE_GROUP);
    }
}
void mgk_scatter(hipStream_t st, const ScatterItem *items, uint32_t nitems) {
    if (!nitems) return;
    uint32_t grid = nitems < 8192 ? nitems : 8192;
    hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(MG_BLOCK), 0, st, items, nitems);
}
// ------------------------------------------------------------------ f-4: highlighting, the matcher
// hit_query_processing (query.c:189-310) highlights the stored strings of every hit on the page: highlight() cuts a string
// into tokens and tries every token against every query term with a prefix-returning Damerau-Levenshtein (is_match /
// levenshtein, highlight.c:166-213).  A token's verdict -- the first term that matches and how many of its leading
// characters the match covers -- depends on the token and the term set alone, so the tokens of all strings of all hits of
// a batch of queries are matched in ONE launch, one lane per token; the tokenizer and the mark assembly (sequential per
// string: overlap rule, snippet window) stay with the caller (marlin_amd/host/mhighlight.c).
// The distance matrix is the reference's full (term + 1) x (token + 1) table, kept as three rows of bytes per lane in
// LDS (values never exceed 254: strings are at most 127 characters).
#define HL_MAXLEN 128
struct HlTerm { uint32_t off, len; int32_t max_dist, prefix; };
__global__ __launch_bounds__(64) void k_hl_match(const uint32_t *tok_chars, const uint32_t *tok_off, const uint32_t *tok_job, uint32_t ntok,
                                                 const HlTerm *terms, const uint32_t *term_chars, const uint32_t *job_term_off, int32_t *out) {
    __shared__ uint8_t rows[3 * HL_MAXLEN * 64];
    const uint32_t lane = threadIdx.x, t = blockIdx.x * 64 + lane;
    if (t >= ntok) return;
    const uint32_t *s1 = tok_chars + tok_off[t];
    const int s1len = (int)(tok_off[t + 1] - tok_off[t]);
    int32_t res = -1;
    if (s1len >= 1 && s1len < HL_MAXLEN) {
        const uint32_t job = tok_job[t];
        for (uint32_t j = job_term_off[job]; j < job_term_off[job + 1] && res < 0; j++) {
            const HlTerm tm = terms[j];
            const int s2len = (int)tm.len, max_dist = tm.max_dist;
            if (s2len < 1 || s2len >= HL_MAXLEN) continue;
            if (!tm.prefix && (s1len < s2len - max_dist || s1len > s2len + max_dist)) continue;          // is_match, highlight.c:199-203
            const uint32_t *s2 = term_chars + tm.off;
#define ROW(r, y) rows[((r) * HL_MAXLEN + (y)) * 64 + lane]
            // row x of the table lives in rows[x % 3]
            for (int y = 0; y <= s1len; y++) ROW(0, y) = (uint8_t)y;
            int best_pos = 0xFF;
            bool dead = false;
            for (int x = 1; x <= s2len && !dead; x++) {
                const int r = x % 3, r1 = (x + 2) % 3, r2 = (x + 1) % 3;
                ROW(r, 0) = (uint8_t)x;
                int best = 0xFF;
                const uint32_t c2 = s2[x - 1], c2p = x > 1 ? s2[x - 2] : 0;
                for (int y = 1; y <= s1len; y++) {
                    const uint32_t c1 = s1[y - 1];
                    const int cost = (c1 == c2) ? 0 : 1;
                    int v = (int)ROW(r1, y) + 1;
                    const int a = (int)ROW(r, y - 1) + 1, b = (int)ROW(r1, y - 1) + cost;
                    if (a < v) v = a;
                    if (b < v) v = b;
                    if (x > 1 && y > 1 && c1 == c2p && s1[y - 2] == c2) {
                        const int tt = (int)ROW(r2, y - 2) + cost;
                        if (tt < v) v = tt;
                    }
                    ROW(r, y) = (uint8_t)v;
                    if (v < best) { best = v; best_pos = y; }
                }
                if (best > max_dist) dead = true;
            }
#undef ROW
            if (!dead && best_pos <= s1len) res = best_pos;
        }
    }
    out[t] = res;
}
void mgk_hl_match(hipStream_t st, const uint32_t *tok_chars, const uint32_t *tok_off, const uint32_t *tok_job, uint32_t ntok,
                  const void *terms, const uint32_t *term_chars, const uint32_t *job_term_off, int32_t *out) {
    if (!ntok) return;
    hipLaunchKernelGGL(k_hl_match, dim3((ntok + 63) / 64), dim3(64), 0, st, tok_chars, tok_off, tok_job, ntok, (const HlTerm *)terms, term_chars, job_term_off, out);
}
// ------------------------------------------------------------------ f-2: mirror maintenance
// A committed bulk write (sindex.c:184-254: si_write_end stores the changed mbmaps, the new docdata blobs and num_dbi
// entries) appends documents: every posting gains docids above the old maximum.  The mirror takes that as a delta
// instead of being rebuilt (mgpu_shard_apply_delta): the sparse posting array is re-laid on the device -- every
// posting's old run followed by its new docids (mbmap_save's effect on the stored containers, mbmap.c:183-251) -- and the
// window index of the long postings is recomputed from the merged lists; dense rows take their new bits through k_scatter.
// One wave per posting.
__global__ __launch_bounds__(MG_BLOCK) void k_merge_postings(uint32_t num_postings, const uint32_t *old_pid, const int32_t *new_dense,
                                                             const u64 *new_off, const int32_t *old_dense, const u64 *old_off, const uint32_t *old_doc,
                                                             const u64 *delta_off, const uint32_t *delta_doc, uint32_t *new_doc) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t p = blockIdx.x * (MG_BLOCK / 64) + (threadIdx.x >> 6); p < num_postings; p += gridDim.x * (MG_BLOCK / 64)) {
        if (new_dense[p] >= 0) continue;
        uint32_t *dst = new_doc + new_off[p];
        u64 n0 = 0;
        const uint32_t op = old_pid[p];
        if (op != 0xFFFFFFFFu && old_dense[op] < 0) {
            const u64 a = old_off[op]; n0 = old_off[op + 1] - a;
            for (u64 i = lane; i < n0; i += 64) dst[i] = old_doc[a + i];
        }
        const u64 a = delta_off[p], n1 = delta_off[p + 1] - a;
        for (u64 i = lane; i < n1; i += 64) dst[n0 + i] = delta_doc[a + i];
    }
}
// post_win[winbase[p] + j] = first element of posting p (relative to its start) whose docid is >= j * 64 * MG_FORMULA_WORDS,
// j = 0 .. nwin (the last entry = the posting's length).  One wave per indexed posting, one lane per window.
__global__ __launch_bounds__(MG_BLOCK) void k_build_winidx(uint32_t num_postings, const uint32_t *winbase, const u64 *off, const uint32_t *doc,
                                                           uint32_t nwin, uint32_t *post_win) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t p = blockIdx.x * (MG_BLOCK / 64) + (threadIdx.x >> 6); p < num_postings; p += gridDim.x * (MG_BLOCK / 64)) {
        const uint32_t wb = winbase[p];
        if (wb == 0xFFFFFFFFu) continue;
        const uint32_t *list = doc + off[p];
        const uint32_t n = (uint32_t)(off[p + 1] - off[p]);
        for (uint32_t j = lane; j <= nwin; j += 64) {
            uint32_t lo = 0, hi = n;
            if (j < nwin) {
                const u64 lim = (u64)j * 64 * MG_FORMULA_WORDS;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((u64)list[mid] < lim) lo = mid + 1; else hi = mid; }
            } else lo = n;
            post_win[wb + j] = lo;
        }
    }
}
void mgk_merge_postings(hipStream_t st, uint32_t num_postings, const uint32_t *old_pid, const int32_t *new_dense, const unsigned long long *new_off,
                        const int32_t *old_dense, const unsigned long long *old_off, const uint32_t *old_doc,
                        const unsigned long long *delta_off, const uint32_t *delta_doc, uint32_t *new_doc,
                        const uint32_t *winbase, uint32_t nwin, uint32_t *post_win) {
    if (!num_postings) return;
    uint32_t g = (num_postings + MG_BLOCK / 64 - 1) / (MG_BLOCK / 64); if (g > 16384) g = 16384;
    hipLaunchKernelGGL(k_merge_postings, dim3(g), dim3(MG_BLOCK), 0, st, num_postings, old_pid, new_dense, (const u64 *)new_off, old_dense,
                       (const u64 *)old_off, old_doc, (const u64 *)delta_off, delta_doc, new_doc);
    hipLaunchKernelGGL(k_build_winidx, dim3(g), dim3(MG_BLOCK), 0, st, num_postings, winbase, (const u64 *)new_off, (const uint32_t *)new_doc, nwin, post_win);
}
// The batch's lowered plans, fetched from pinned host memory by the GPU itself (16 bytes per lane over PCIe) instead
// of a hipMemcpyAsync: on this runtime an H2D copy enqueued behind kernels of other streams was seen to hold the calling
// host thread for milliseconds (profiles/README.md, round 2), a kernel launch returns at once.
__global__ __launch_bounds__(MG_BLOCK) void k_fetch(uint4 *dst, const uint4 *src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * MG_BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * MG_BLOCK) dst[i] = src[i];
}
void mgk_fetch(hipStream_t st, void *dst, const void *src_pinned, size_t bytes) {
    const size_t n16 = (bytes + 15) / 16;
    if (!n16) return;
    size_t g = (n16 + MG_BLOCK - 1) / MG_BLOCK; if (g > 128) g = 128;
    hipLaunchKernelGGL(k_fetch, dim3((uint32_t)g), dim3(MG_BLOCK), 0, st, (uint4 *)dst, (const uint4 *)src_pinned, n16);
}
void mgk_scatter_list(hipStream_t st, const uint32_t *src, uint64_t n, unsigned long long *dst) {
    if (!n) return;
    const uint64_t nchunks = (n + MG_SCATTER_CHUNK - 1) / MG_SCATTER_CHUNK;
    hipLaunchKernelGGL(k_scatter_list, dim3((uint32_t)(nchunks < 8192 ? nchunks : 8192)), dim3(MG_BLOCK), 0, st, src, n, (u64 *)dst);
}
// number of workgroups of the persistent kernels that stride over the tile work list
static uint32_t worklist_grid(uint32_t tile_cap) {
    uint32_t g = 256 * 8;                                  // 256 CUs, a few workgroups each
    if (g > tile_cap) g = tile_cap;
    return g ? g : 1;
}
// squeries [0, nmain) are the batch's own, [nmain, nsq) the zero-typo twins of those that may be ranked on a sample: the
// twins run in a second pair of launches, after k_order has decided which squeries are sampled, and the others' return at once
// (one squery in thirty of the default-mode bench stream is sampled; its twins used to double the formula stage)
void mgk_formula(hipStream_t st, const SqDev *sqs, const uint32_t *blob, uint32_t max_nblk, uint32_t nsq, uint32_t nmain,
                 uint2 *tile_list, uint32_t *tile_count, uint32_t tile_cap, uint32_t *fh_list, uint32_t *fh_count, uint32_t fh_cap) {
    // runs of windows pay when the launch fills the chip several times over (-12 % on a 1024-squery batch); a lone query's
    // few hundred windows are better off side by side
    const uint32_t max_nwin = (max_nblk + 1) / 2;
    static const int forced = getenv("MGPU_FORMULA_RUN") ? atoi(getenv("MGPU_FORMULA_RUN")) : 0;     // tests: runs on small shards
    if (nmain > nsq) nmain = nsq;
    for (int part = 0; part < 2; part++) {
        const uint32_t first = part ? nmain : 0u, cnt = part ? nsq - nmain : nmain;
        if (!cnt) continue;
        const uint32_t run = forced > 0 ? (uint32_t)forced : (uint64_t)max_nwin * cnt >= 8192 ? MG_FORMULA_RUN : 1u;
        hipLaunchKernelGGL(k_formula, dim3((max_nwin + run - 1) / run, cnt), dim3(MG_BLOCK), 0, st, sqs + first, blob, run);
        hipLaunchKernelGGL(k_order, dim3(cnt), dim3(MG_BLOCK), 0, st, sqs + first, tile_list, tile_count, fh_list, fh_count, fh_cap);
    }
    hipLaunchKernelGGL(k_order_copy, dim3(worklist_grid(tile_cap)), dim3(MG_BLOCK), 0, st, sqs, tile_list, tile_count);
}
void mgk_fast_build(hipStream_t st, const SqDev *sqs, uint32_t nsq, const uint32_t *elig_list, uint32_t nelig,
                    uint2 *tile_list_b, uint32_t *tile_count_b) {
    if (!nelig) return;
    hipLaunchKernelGGL(k_fast_build, dim3(nelig), dim3(FB_NT), 0, st, sqs, elig_list);
    hipLaunchKernelGGL(k_tiles_again, dim3((nsq + MG_BLOCK - 1) / MG_BLOCK), dim3(MG_BLOCK), 0, st, sqs, nsq, tile_list_b, tile_count_b);
}
void mgk_facet_recount(hipStream_t st, const SqDev *sqs, const void *jobs, uint32_t njobs, uint32_t max_k) {
    if (!njobs) return;
    const uint32_t gy = max_k < 1 ? 1u : max_k > 64 ? 64u : max_k;         // (longer lists stride)
    hipLaunchKernelGGL(k_facet_recount, dim3(njobs, gy), dim3(MG_BLOCK), 0, st, sqs, (const RecountJob *)jobs);
}
void mgk_rank(hipStream_t st, const SqDev *sqs, const uint32_t *blob, uint32_t max_n, uint32_t nsq,
              uint32_t max_tab_slots, int with_len, const uint2 *tile_list, const uint32_t *tile_count, uint32_t tile_cap,
              unsigned long long *pool, uint32_t *pool_used, uint64_t pool_cap) {
    if (max_tab_slots > MG_LDS_HASH) max_tab_slots = MG_LDS_HASH;
    // workgroups of the persistent grid: MGPU_RANK_GRID overrides the default (a tuning knob; any value >= 1 is correct)
    static const uint32_t grid_env = [] { const char *e = getenv("MGPU_RANK_GRID"); long v = e ? atol(e) : 0; return (uint32_t)(v > 0 && v <= 65535 ? v : 0); }();
    uint32_t grid = grid_env ? grid_env : worklist_grid(tile_cap);
    if (grid > tile_cap && tile_cap) grid = tile_cap;
    if (with_len) hipLaunchKernelGGL(k_rank_tile<true>, dim3(grid), dim3(MG_BLOCK), (size_t)max_tab_slots * sizeof(uint2), st, sqs, blob, tile_list, tile_count);
    else hipLaunchKernelGGL(k_rank_tile<false>, dim3(grid), dim3(MG_BLOCK), (size_t)max_tab_slots * sizeof(uint2), st, sqs, blob, tile_list, tile_count);
    uint32_t gy = (max_n + MG_BLOCK - 1) / MG_BLOCK;
    uint32_t cap = 4096 / nsq; if (cap < 4) cap = 4;
    if (gy > cap) gy = cap;
    hipLaunchKernelGGL(k_rank_overflow, dim3(gy, nsq), dim3(MG_BLOCK), 0, st, sqs, blob, pool, pool_used, pool_cap);
}
// lds_bins: the largest histogram of the batch that has at most MG_LDS_HIST bins (those are privatised in LDS;
// sizing the LDS by it instead of by the limit keeps more workgroups per CU when the small fields are small)
void mgk_docproc(hipStream_t st, const SqDev *sqs, uint32_t nsq, const uint2 *tile_list, const uint32_t *tile_count,
                 uint32_t tile_cap, uint32_t lds_bins, int pre) {
    if (lds_bins > MG_LDS_HIST) lds_bins = MG_LDS_HIST;
    hipLaunchKernelGGL(k_docproc, dim3(worklist_grid(tile_cap)), dim3(MG_BLOCK), (size_t)lds_bins * 4, st, sqs, tile_list, tile_count, lds_bins, pre);
    hipLaunchKernelGGL(k_aggfinal, dim3(nsq), dim3(64), 0, st, sqs, pre);
}
void mgk_fkeys(hipStream_t st, const void *jobs, uint32_t njobs, uint32_t max_n) {
    if (!njobs) return;
    uint32_t gx = (max_n + MG_BLOCK - 1) / MG_BLOCK, cap = 8192 / njobs; if (cap < 4) cap = 4;
    if (gx > cap) gx = cap; if (!gx) gx = 1;
    hipLaunchKernelGGL(k_fkeys, dim3(gx, njobs), dim3(MG_BLOCK), 0, st, (const FkJob *)jobs);
}
void mgk_card(hipStream_t st, const void *jobs, uint32_t njobs, uint32_t max_n) {
    if (!njobs) return;
    uint32_t gx = (max_n + MG_BLOCK - 1) / MG_BLOCK, cap = 8192 / njobs; if (cap < 4) cap = 4;
    if (gx > cap) gx = cap; if (!gx) gx = 1;
    hipLaunchKernelGGL(k_card, dim3(gx, njobs), dim3(MG_BLOCK), 0, st, (const CardJob *)jobs);
}
// `finals` = 0: only the pass launches (jobs that fit one workgroup finish inside them);
// 1: also the stand-alone final sorts, for the continuation rounds of larger jobs.
// lists: device memory for the select work lists: counters = 6 zeroed u32 words (4 item counters, the
// final-list counter, the mid-list counter), items = (SEL_WIDE_LAUNCHES + 1) arrays of cap_items uint2,
// fin_list = 2 * (njobs + 1) u32 (final list, then mid list).
// list / count: the two size classes k_order filled ([0 .. cap) small, [cap .. 2 cap) large)
void mgk_facet_hash(hipStream_t st, const SelJob *jobs, const void *outs, const uint32_t *list, const uint32_t *count, uint32_t cap) {
    if (!cap) return;
    hipLaunchKernelGGL((k_facet_hash<MG_BLOCK>), dim3(cap < 768 ? cap : 768), dim3(MG_BLOCK), 0, st, jobs, (const SelOut *)outs, list, count, cap);
    hipLaunchKernelGGL((k_facet_hash<1024>), dim3(cap < 256 ? cap : 256), dim3(1024), 0, st, jobs, (const SelOut *)outs, list + cap, count + 1, cap);
}
void mgk_select(hipStream_t st, const SelJob *jobs, const void *outs, uint32_t njobs, uint32_t max_m, int passes, int finals,
                void *items_mem, uint32_t cap_items, uint32_t *counters, uint32_t *fin_list) {
    if (!njobs) return;
    const SelOut *so = (const SelOut *)outs;
    SelLists L;
    for (int i = 0; i <= SEL_WIDE_LAUNCHES; i++) L.items[i] = (uint2 *)items_mem + (size_t)i * cap_items;
    L.nitems = counters; L.nfin = counters ? counters + 4 : nullptr; L.fin = fin_list; L.cap_items = cap_items;
    L.nmid = counters ? counters + 5 : nullptr; L.mid = fin_list ? fin_list + njobs + 1 : nullptr;
    static const int lanemin = getenv("MGPU_NO_LANEMIN") ? 0 : 1;        // diagnostic switch: the radix select_solo instead
    L.lanemin = lanemin;
    uint32_t gx = (max_m + SEL_CHUNK - 1) / SEL_CHUNK;
    if (!finals) {
        // first go: exactly one workgroup per job, no idle ones ...
        hipLaunchKernelGGL(k_select_first, dim3(njobs), dim3(MG_BLOCK), 0, st, jobs, so, L);
        if (max_m <= MG_SEL_SMALL) return;                                 // no job of this batch can end up on a list
        if (max_m > SEL_SOLO_CHUNKS * SEL_CHUNK) {
            hipLaunchKernelGGL(k_fkeys_mid, dim3(SEL_FKM_WG, njobs < 128 ? njobs : 128), dim3(MG_BLOCK), 0, st, jobs, L);
            hipLaunchKernelGGL(k_select_mid, dim3(njobs < 256 ? njobs : 256), dim3(SEL_MID_BLOCK), 0, st, jobs, so, L);
        }
        // ... then the wide launches over exactly the workgroups the long lists asked for, and the
        // stand-alone sorts for the jobs on the final list
        // (cap_items == 0: the host knows that no job of the batch is longer than one mid workgroup's list or
        // wants more than MG_SEL_SOLO_K keys, so nothing lands on the wide lists or the final list; should a
        // job do so all the same, its out_n stays "not finished" and the host-driven rounds pick it up)
        if (!cap_items) return;
        {
            uint32_t g = cap_items < 2048 ? cap_items : 2048;
            for (int p = 0; p < SEL_WIDE_LAUNCHES; p++)
                hipLaunchKernelGGL(k_select_items, dim3(g), dim3(MG_BLOCK), 0, st, jobs, so, L, p);
        }
        uint32_t fy = njobs < 256 ? njobs : 256;
        hipLaunchKernelGGL((k_select_final<SEL_SMALL_CAP, 0>), dim3(fy), dim3(MG_BLOCK), 0, st, jobs, so, L.fin, L.nfin);
        hipLaunchKernelGGL((k_select_final<SEL_BIG_CAP, SEL_SMALL_CAP + 1>), dim3(fy), dim3(MG_BLOCK), 0, st, jobs, so, L.fin, L.nfin);
        return;
    }
    // host-driven continuation (rare: more than SEL_WIDE_LAUNCHES wide passes needed, or a list overflowed)
    uint32_t cap = 4096 / njobs; if (cap < 4) cap = 4;
    if (gx <= SEL_SOLO_CHUNKS) gx = 1; else if (gx > cap) gx = cap;
    for (int p = 0; p < passes; p++)
        hipLaunchKernelGGL(k_select_pass, dim3(gx, njobs), dim3(MG_BLOCK), 0, st, jobs, so, (int)SELMODE_HOST, L);
    hipLaunchKernelGGL((k_select_final<SEL_SMALL_CAP, 0>), dim3(njobs), dim3(MG_BLOCK), 0, st, jobs, so, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    hipLaunchKernelGGL((k_select_final<SEL_BIG_CAP, SEL_SMALL_CAP + 1>), dim3(njobs), dim3(MG_BLOCK), 0, st, jobs, so, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
}
// test aid (MGPU_CHECK_CLEAN): *flag |= 1 when any word of the self-cleaned pool is not zero
__global__ __launch_bounds__(MG_BLOCK) void k_check_zero(const uint4 *p, size_t n16, uint32_t *flag) {
    uint32_t any = 0;
    for (size_t i = (size_t)blockIdx.x * MG_BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * MG_BLOCK) {
        const uint4 v = p[i];
        any |= v.x | v.y | v.z | v.w;
    }
    if (any) atomicOr(flag, 1u);
}
void mgk_check_zero(hipStream_t st, const void *p, size_t bytes, uint32_t *flag) {
    hipLaunchKernelGGL(k_check_zero, dim3(2048), dim3(MG_BLOCK), 0, st, (const uint4 *)p, bytes / 16, flag);
}
void mgk_row_op(hipStream_t st, const unsigned long long *a, const unsigned long long *b, unsigned long long *out,
                uint32_t W, int op, unsigned long long *card) {
    uint32_t gx = (W + MG_BLOCK - 1) / MG_BLOCK; if (gx > 2048) gx = 2048; if (!gx) gx = 1;
    hipLaunchKernelGGL(k_row_op, dim3(gx), dim3(MG_BLOCK), 0, st, a, b, out, W, op, card);
}
}
