// bf16-feature sparse propagation  Y = act(adj @ X)  for BASELINE configs[4] (GraphConvolution.forward's
// `torch.matmul(adj, support)`, models/Multi_GCN_Multihead_att.py:52-58, with the gen_adj'd adjacency of
// utils/util.py:421-426 held sparse): adjacency values bf16, X bf16 [n_cols, F], fp32 accumulation in ascending
// column order, Y bf16 or fp32.  Algorithmic bytes: nnz * (4 + 2) + n * F * 2 (X) + n * F * 2 (Y)  (SURVEY 7-8).
//
// Two kernels, picked by the caller from the graph's density:
//   * spmm_bf16_slab_kernel   -- PMI-like graphs (a few non-zeros per row): every non-zero is ONE gather of a row
//     segment of X straight from the XCD's L2.  The feature axis is cut into 128-feature slabs (256 B of a row) owned by
//     XCDs, so that the slab of X an XCD works on (n * 256 B = 2.5 MB at 10 000 nodes) stays in its 4-MiB L2 and X
//     crosses the fabric once.  HBM-bound: one read of X, one write of Y.
//   * spmm_bf16_tiled_kernel  -- dense-ish graphs (tens of non-zeros per row and more): the L2 cannot feed one gather per
//     non-zero (1e6 non-zeros x 2 KB = 2 GB per product at density 1e-2), so X is staged through LDS: a workgroup owns
//     R rows x FS features of Y in REGISTERS and marches over column blocks; the X tile [BC columns x FS features] of the
//     current block arrives by LDS-DMA (double buffered) and every non-zero of the block becomes one conflict-free LDS
//     row read (64 lanes x 8 B) + dot products.  Needs the adjacency re-ordered once into per-(wave, column block) entry
//     streams (mgnns_amd/spmm_plan.py; the adjacency is a static parameter of the model).
#include "common.hpp"

namespace {

typedef __bf16 sb_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int sb_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned sb_pack_bf16(float a, float b) {   // (lo = a, hi = b), round to nearest even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ------------------------------------------------------------------------------------------------------------
// Direct (gather) kernel.  A lane group of 16 lanes owns one row at a time: 16 lanes x 16 B = the 256-B segment of one
// 128-feature slab.  RU rows per lane group and NS adjacent slabs are in flight together (ONE fetch of the row's col / val
// for all NS slabs), up to four non-zeros per row at a time: 4 * RU * NS independent 16-B gathers per lane.
// blockIdx & 7 = XCD (observed dispatch order; only speed depends on it): XCD x owns slabs [x * spx, (x + 1) * spx).
template <int NS, int RU>
__global__ __launch_bounds__(256) void spmm_bf16_slab_kernel(const int32_t* __restrict__ row_ptr,
                                                             const int32_t* __restrict__ col,
                                                             const uint16_t* __restrict__ val,
                                                             const uint16_t* __restrict__ X, int n_rows, int F,
                                                             void* __restrict__ Yv, int y_bf16, int act,
                                                             const int32_t* __restrict__ row_map) {
    constexpr int SW = 128;                     // slab width in features (16 lanes x 8 bf16)
    const int xcd = blockIdx.x & 7, bj = blockIdx.x >> 3, nb = gridDim.x >> 3;
    const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15;
    const int wave = bj * 4 + (threadIdx.x >> 6), nwaves = nb * 4;
    const int nslabs = (F + SW - 1) / SW;
    const int spx = (nslabs + 7) / 8;
    for (int slab0 = xcd * spx; slab0 < (xcd + 1) * spx && slab0 < nslabs; slab0 += NS) {
        int f[NS];
        bool fon[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int sl = slab0 + q;
            f[q] = sl * SW + l * 8;
            fon[q] = sl < nslabs && sl < (xcd + 1) * spx && f[q] < F;
        }
        for (int r0 = wave * 4 * RU; r0 < n_rows; r0 += nwaves * 4 * RU) {
            int p[RU], hi[RU], row[RU], orow[RU];      // row: position in the (possibly length-sorted) CSR, orow: the row of Y it is
            float acc[RU][NS][8];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                row[u] = r0 + 4 * u + g;
                orow[u] = row[u];
                p[u] = hi[u] = 0;
#pragma unroll
                for (int q = 0; q < NS; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[u][q][j] = 0.f;
                if (row[u] < n_rows) {
                    p[u] = row_ptr[row[u]];
                    hi[u] = row_ptr[row[u] + 1];
                    if (row_map) orow[u] = row_map[row[u]];
                }
            }
            bool more = false;
#pragma unroll
            for (int u = 0; u < RU; ++u) more |= p[u] < hi[u];
            while (__any(more)) {
                int c[RU][4];
                float w[RU][4];
                u32x4 x[RU][4][NS];
#pragma unroll
                for (int u = 0; u < RU; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool live = p[u] + k < hi[u];
                        c[u][k] = live ? col[p[u] + k] : 0;
                        w[u][k] = live ? __builtin_bit_cast(float, (unsigned)val[p[u] + k] << 16) : 0.f;
                    }
#pragma unroll
                for (int u = 0; u < RU; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int q = 0; q < NS; ++q) {
                            x[u][k][q] = u32x4{0u, 0u, 0u, 0u};
                            if (p[u] + k < hi[u] && fon[q])
                                x[u][k][q] = *reinterpret_cast<const u32x4*>(X + (size_t)c[u][k] * F + f[q]);
                        }
                more = false;
#pragma unroll
                for (int u = 0; u < RU; ++u) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int q = 0; q < NS; ++q)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const unsigned d = x[u][k][q][j];
                                acc[u][q][2 * j] = fmaf(w[u][k], __builtin_bit_cast(float, d << 16), acc[u][q][2 * j]);
                                acc[u][q][2 * j + 1] = fmaf(w[u][k], __builtin_bit_cast(float, d & 0xffff0000u), acc[u][q][2 * j + 1]);
                            }
                    p[u] += 4;
                    more |= p[u] < hi[u];
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u)
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    if (fon[q] && row[u] < n_rows) {
                        float o[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = mg_act(acc[u][q][j], act);
                        if (y_bf16) {
                            const u32x4 ov = {sb_pack_bf16(o[0], o[1]), sb_pack_bf16(o[2], o[3]), sb_pack_bf16(o[4], o[5]),
                                              sb_pack_bf16(o[6], o[7])};
                            __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(static_cast<uint16_t*>(Yv) + (size_t)orow[u] * F + f[q]));
                        } else {
                            float* yp = static_cast<float*>(Yv) + (size_t)orow[u] * F + f[q];
                            __builtin_nontemporal_store(f32x4{o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4*>(yp));
                            __builtin_nontemporal_store(f32x4{o[4], o[5], o[6], o[7]}, reinterpret_cast<f32x4*>(yp + 4));
                        }
                    }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Direct kernel, ring form: the gathers land in LDS by LDS-DMA (no VGPR per byte in flight) and never stop.  A wave owns a
// CONTIGUOUS range of rows and streams it: its non-zeros, in CSR order, map onto a ring of RI DMA pieces (1 KiB = 4 / NSL row
// segments each); the issue pointer runs up to the ring's capacity ahead of the rows being reduced; col / val travel through
// LDS rings of four 64-entry windows (fetched by LDS-DMA as well, one window ahead), row pointers in a 64-row register window.  Vector-memory operations complete in order, so "piece k has landed" is
// s_waitcnt vmcnt(pieces issued after k) -- an immediate, hence the switch.  Stores issued in between are ignored by that
// count, which only makes the wait conservative.  Rows are reduced four at a time (one per 16-lane group), in ascending
// column order: bit-identical to the other two forms.
__device__ __forceinline__ void sb_wait_vm(int n) {
#define MG_W(k) \
    case k:     \
        asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); \
        break;
    switch (n) {
        MG_W(0) MG_W(1) MG_W(2) MG_W(3) MG_W(4) MG_W(5) MG_W(6) MG_W(7) MG_W(8) MG_W(9) MG_W(10) MG_W(11) MG_W(12) MG_W(13)
        MG_W(14) MG_W(15) MG_W(16) MG_W(17) MG_W(18) MG_W(19) MG_W(20) MG_W(21) MG_W(22) MG_W(23) MG_W(24) MG_W(25) MG_W(26)
        MG_W(27) MG_W(28) MG_W(29) MG_W(30) MG_W(31)
        default:
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    }
#undef MG_W
}
__device__ __forceinline__ unsigned sb_lds_read32(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ unsigned sb_lds_read16(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

template <int NSL, int RI>
__global__ __launch_bounds__(256) void spmm_bf16_ring_kernel(const int32_t* __restrict__ row_ptr,
                                                             const int32_t* __restrict__ col,
                                                             const uint16_t* __restrict__ val,
                                                             const uint16_t* __restrict__ X, int n_rows, int F,
                                                             void* __restrict__ Yv, int y_bf16, int act, int nnz,
                                                             const int32_t* __restrict__ row_map) {
    constexpr int SB = NSL * 256, LPS = SB / 16, EPI = 64 / LPS, RING_E = RI * EPI;
    constexpr int WAVE_LDS = RI * 1024 + 1024 + 512;               // gather ring + col ring (256 x 4 B) + val ring (256 x 2 B)
    static_assert(RING_E <= 64, "the gather ring must not outrun the four 64-entry metadata windows");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char* wbase = smem + wv * WAVE_LDS;
    unsigned char* colring = wbase + RI * 1024;
    unsigned char* valring = colring + 1024;
    const unsigned wbase_a = mg_lds_addr(wbase), col_a = mg_lds_addr(colring), val_a = mg_lds_addr(valring);
    const int xcd = blockIdx.x & 7, bj = blockIdx.x >> 3, nb = gridDim.x >> 3;
    const int wave_x = bj * 4 + wv, nwaves_x = nb * 4;
    const int nslabs = (F + 127) / 128;
    const int spx = (nslabs + 7) / 8;
    const int slab_end = min((xcd + 1) * spx, nslabs);
    const int rows_per_wave = ((n_rows + nwaves_x - 1) / nwaves_x + 3) & ~3;
    const int r_begin = wave_x * rows_per_wave, r_end = min(n_rows, r_begin + rows_per_wave);
    if (r_begin >= r_end) return;
    const int e0 = __builtin_amdgcn_readfirstlane(row_ptr[r_begin]);
    const int total_e = __builtin_amdgcn_readfirstlane(row_ptr[r_end]) - e0;
    const int total_i = (total_e + EPI - 1) / EPI;
    // Metadata windows live in SHIFTED entry coordinates k' = k + pe (pe = parity of the wave's first entry), so that the bf16
    // values can be fetched as aligned dwords: window w = entries k' in [64 w, 64 w + 64), DMA'd into slot w & 3 of the col ring
    // (4 B per entry) and of the val ring (2 B per entry).  val must be readable up to an even number of elements.
    const int pe = e0 & 1, a0 = e0 - pe;
    const int n_win = (total_e + pe + 63) / 64;
    auto request_window = [&](int w) {
        if (a0 + 64 * w + lane < nnz)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(col + a0 + 64 * w + lane),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(colring + (w & 3) * 256), 4, 0, 0);
        if (lane < 32 && a0 + 64 * w + 2 * lane < nnz)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(val + a0 + 64 * w + 2 * lane),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(valring + (w & 3) * 128), 4, 0, 0);
    };
    for (int slab0 = xcd * spx; slab0 < slab_end; slab0 += NSL) {
        const int fo = slab0 * 128 + (lane % LPS) * 8;            // DMA lane -> feature offset inside the gathered row
        const bool fo_on = fo < F && (slab0 + (lane % LPS) / 16) < slab_end;
        int f[NSL];
        bool fon[NSL];
#pragma unroll
        for (int q = 0; q < NSL; ++q) {
            f[q] = (slab0 + q) * 128 + l * 8;
            fon[q] = slab0 + q < slab_end && f[q] < F;
        }
        int rwin = r_begin;                                        // first row of the row-pointer window
        int rpv = row_ptr[min(rwin + lane, n_rows)], rpe = row_ptr[min(rwin + lane + 1, n_rows)];
        int rmv = row_map ? row_map[min(rwin + lane, n_rows - 1)] : rwin + lane;      // the row of Y behind CSR position rwin + lane
        asm volatile("" : "+v"(rpv), "+v"(rpe), "+v"(rmv));        // landed here (nothing else is in flight yet)
        request_window(0);
        request_window(1);
        int wreq = 1;                                              // highest window requested
        int since_req = 0;                                         // DMA pieces issued after that request
        int wiss = -1;                                             // highest window the issue pointer has entered (and waited for)
        int issued = 0;                                            // DMA pieces issued
        int free_e = 0;                                            // entries below this are reduced (their ring slots are free)
        for (int r = r_begin; r < r_end; r += 4) {
            if (r - rwin >= 64) {                                  // next 64 rows (drains the pipe; once per 64 rows)
                rwin = r;
                rpv = row_ptr[min(rwin + lane, n_rows)];
                rpe = row_ptr[min(rwin + lane + 1, n_rows)];
                rmv = row_map ? row_map[min(rwin + lane, n_rows - 1)] : rwin + lane;
                asm volatile("" : "+v"(rpv), "+v"(rpe), "+v"(rmv));
            }
            const int rr = r + g;
            const int ro = __shfl(rmv, min(r - rwin + g, 63), 64);  // where this lane group's row goes
            const int rs = __shfl(rpv, r - rwin + g, 64) - e0, re = __shfl(rpe, r - rwin + g, 64) - e0;   // this lane group's row
            const int chunk_end = __builtin_amdgcn_readlane(rpe, min(r + 3, r_end - 1) - rwin) - e0;
            float acc[NSL][8];
#pragma unroll
            for (int q = 0; q < NSL; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
            for (;;) {
                // top up the gather ring, up to four pieces per round trip to the col ring
                for (;;) {
                    const int room = min(total_i, (free_e + RING_E) / EPI) - issued;
                    if (room <= 0) break;
                    const int cnt = min(room, 4);
                    const int w = ((issued + cnt) * EPI - 1 + pe) >> 6;
                    if (w > wiss) {                                // the issue pointer enters window w: it must have landed
                        sb_wait_vm(since_req);
                        wiss = w;
                        if (w + 1 > wreq && w + 1 < n_win) {       // ... and window w + 1 is requested (its slot held window w - 3)
                            request_window(w + 1);
                            wreq = w + 1;
                            since_req = 0;
                        }
                    }
                    int cc[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) cc[p] = (int)sb_lds_read32(col_a + (unsigned)((((issued + p) * EPI + lane / LPS + pe) & 255) * 4));
                    mg_lds_wait<0>();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        if (p < cnt) {
                            if ((issued + p) * EPI + lane / LPS < total_e && fo_on)
                                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (size_t)cc[p] * F + fo),
                                                                 (__attribute__((address_space(3))) void*)(uintptr_t)(wbase + ((issued + p) % RI) * 1024), 16, 0, 0);
                        }
                    issued += cnt;
                    since_req += cnt;
                }
                const int need_i = min(issued, (chunk_end + EPI - 1) / EPI);
                sb_wait_vm(issued - need_i);
                const int landed = min(need_i * EPI, total_e);
                // reduce this lane group's row over [max(rs, free_e), min(re, landed)), up to eight entries per LDS round trip
                int sidx = max(rs, free_e);
                const int send = rr < r_end ? min(re, landed) : 0;
                for (;;) {
                    int nmax = max(send - sidx, 0);                // longest remainder among the four rows (wave-uniform)
                    nmax = max(nmax, __shfl_xor(nmax, 16, 64));
                    nmax = max(nmax, __shfl_xor(nmax, 32, 64));
                    nmax = __builtin_amdgcn_readfirstlane(nmax);
                    if (nmax <= 0) break;
                    unsigned wb[8];
                    u32x4 x[8][NSL];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k < nmax) {
                            const int sl = sidx + k < send ? sidx + k : landed - 1;   // idle lanes read some landed entry (finite)
                            wb[k] = sb_lds_read16(val_a + (unsigned)(((sl + pe) & 255) * 2));
#pragma unroll
                            for (int q = 0; q < NSL; ++q) x[k][q] = mg_lds_read128<0>(wbase_a + (unsigned)((sl % RING_E) * SB + q * 256 + l * 16));
                        }
                    mg_lds_wait<0>();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k < nmax) {
                            const float w = sidx + k < send ? __builtin_bit_cast(float, wb[k] << 16) : 0.f;
#pragma unroll
                            for (int q = 0; q < NSL; ++q)
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    acc[q][2 * j] = fmaf(w, __builtin_bit_cast(float, x[k][q][j] << 16), acc[q][2 * j]);
                                    acc[q][2 * j + 1] = fmaf(w, __builtin_bit_cast(float, x[k][q][j] & 0xffff0000u), acc[q][2 * j + 1]);
                                }
                        }
                    sidx += 8;
                }
                free_e = min(landed, chunk_end);
                if (landed >= chunk_end) break;
            }
            if (rr < r_end) {
#pragma unroll
                for (int q = 0; q < NSL; ++q)
                    if (fon[q]) {
                        float o[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = mg_act(acc[q][j], act);
                        if (y_bf16) {
                            const u32x4 ov = {sb_pack_bf16(o[0], o[1]), sb_pack_bf16(o[2], o[3]), sb_pack_bf16(o[4], o[5]),
                                              sb_pack_bf16(o[6], o[7])};
                            __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(static_cast<uint16_t*>(Yv) + (size_t)ro * F + f[q]));
                        } else {
                            float* yp = static_cast<float*>(Yv) + (size_t)ro * F + f[q];
                            __builtin_nontemporal_store(f32x4{o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4*>(yp));
                            __builtin_nontemporal_store(f32x4{o[4], o[5], o[6], o[7]}, reinterpret_cast<f32x4*>(yp + 4));
                        }
                    }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the rings are reused by the next slab pass
    }
}

// ------------------------------------------------------------------------------------------------------------
// LDS-tiled kernel.  Geometry (must match the plan, mgnns_amd/spmm_plan.py):
//   16 waves per workgroup, U rows per wave (R = 16 U rows per workgroup), lane = LB bytes = LB / 2 features of a row,
//   FS = 32 LB features per workgroup (LB = 8: 256 features, one 512-B LDS row per column), BC columns per tile,
//   tile = BC * 64 * LB bytes <= 64 KB, two tiles in LDS.
// Plan: wave_off[(rb * 16 + w) * (ncb + 1) + cb] = dword offset in `ent` of the record of (row block rb, wave w, column
//   block cb).  A record is one or more PARTS; a part = header dword (entries in the part | last-part flag << 31), HD =
//   ceil(U / 4) dwords of per-row entry counts (one byte each), then the entries in (row, column) order, at most
//   64 - 1 - HD per part so that a part is ONE coalesced 256-B load.  Entry = (LDS byte offset of the column's tile row) << 16 |
//   bf16 value.  Entries reach the scalar unit through v_readlane; everything that indexes an accumulator is compile-time.
// Accumulation: D += w * x as v_dot2c_f32_bf16 with the weight paired with a zero ((w, 0) / (0, w)): exact products,
// ascending columns inside a row like the direct kernels (the instruction's own rounding: results agree with their fmaf
// chain to the last bit or two, measured <= 1e-6 of the output scale over 100 non-zeros).  X must be finite: the zero
// partner multiplies the neighbouring feature.
template <int LB>
struct sb_xreg;
template <>
struct sb_xreg<8> {
    typedef sb_u32x2 type;
};
template <>
struct sb_xreg<4> {
    typedef unsigned type;
};

template <int LB>
__device__ __forceinline__ typename sb_xreg<LB>::type sb_lds_read(unsigned addr);
template <>
__device__ __forceinline__ sb_u32x2 sb_lds_read<8>(unsigned addr) {
    sb_u32x2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <>
__device__ __forceinline__ unsigned sb_lds_read<4>(unsigned addr) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

__device__ __forceinline__ float sb_dot(unsigned x, unsigned w, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sb_bf16x2, x), __builtin_bit_cast(sb_bf16x2, w), acc, false);
}
template <int LB>
__device__ __forceinline__ void sb_fma(float* acc, typename sb_xreg<LB>::type x, unsigned e);
template <>
__device__ __forceinline__ void sb_fma<8>(float* acc, sb_u32x2 x, unsigned e) {
    const unsigned wlo = e & 0xffffu, whi = e << 16;
    acc[0] = sb_dot(x[0], wlo, acc[0]);
    acc[1] = sb_dot(x[0], whi, acc[1]);
    acc[2] = sb_dot(x[1], wlo, acc[2]);
    acc[3] = sb_dot(x[1], whi, acc[3]);
}
template <>
__device__ __forceinline__ void sb_fma<4>(float* acc, unsigned x, unsigned e) {
    const unsigned wlo = e & 0xffffu, whi = e << 16;
    acc[0] = sb_dot(x, wlo, acc[0]);
    acc[1] = sb_dot(x, whi, acc[1]);
}

template <int LB, int U, int BC>
__global__ __launch_bounds__(1024) void spmm_bf16_tiled_kernel(const uint32_t* __restrict__ wave_off,
                                                               const uint32_t* __restrict__ ent,
                                                               const uint16_t* __restrict__ X, int n_rows, int n_cols,
                                                               int F, void* __restrict__ Yv, int y_bf16, int act,
                                                               int n_rb, int ncb) {
    constexpr int NW = 16, ROWB = 64 * LB, FS = ROWB / 2, TILEB = BC * ROWB, HD = (U + 3) / 4, NA = LB / 2;
    constexpr int RPP = 1024 / ROWB;                     // tile rows per 1-KiB DMA piece
    constexpr int PIECES = TILEB / 1024, PPW = PIECES / NW;
    static_assert(TILEB <= 65536 && PIECES % NW == 0, "tile geometry");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    typedef typename sb_xreg<LB>::type xreg;

    // workgroup -> (slab, row block): slab-major ranges per XCD, so the workgroups that share the X tiles of a slab sit on
    // one (or two) XCDs and the tile crosses the fabric once per XCD
    const int nsl = F / FS, total = nsl * n_rb;
    const int per_xcd = (total + 7) / 8;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || t >= total) return;
    const int slab = t / n_rb, rb = t % n_rb;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t* wo = wave_off + (size_t)(rb * NW + wave) * (ncb + 1);

    float acc[U][NA];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[u][a] = 0.f;

    // DMA piece p of a tile = tile rows [p * RPP, (p + 1) * RPP); lane -> (row in piece, 16-B chunk)
    const int prow = lane / (ROWB / 16), pchunk = lane % (ROWB / 16);
    const uint16_t* xslab = X + (size_t)slab * FS + pchunk * 8;
    auto issue_tile = [&](int cb, int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + NW * i;
            int c = cb * BC + p * RPP + prow;
            c = c < n_cols ? c : n_cols - 1;                      // columns past the end: never referenced by an entry
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xslab + (size_t)c * F),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(smem + buf * TILEB + p * 1024), 16, 0, 0);
        }
    };

    unsigned voff = wo[lane < ncb + 1 ? lane : ncb];                       // record offsets of column blocks 0..63
    issue_tile(0, 0);
    unsigned vrec = ent[__builtin_amdgcn_readlane(voff, 0) + lane];
    // a compiler-visible use: its own wait for the first record lands HERE.  Otherwise the record counts as in flight on
    // loop entry (the asm waits below are invisible to the wait-count pass) and every iteration opens with vmcnt(0), i.e.
    // waits for the tile just requested.
    asm volatile("" : "+v"(vrec));
    for (int cb = 0; cb < ncb; ++cb) {
        // tile cb has landed for every wave and every wave is done with tile cb - 1
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        unsigned vnext = 0;
        if (cb + 1 < ncb) {
            // order matters: vector-memory results return in order, so the record of the next column block is requested
            // BEFORE the tile (the wait for it is then vmcnt(PPW), and the tile stays in flight behind the whole block)
            if (((cb + 1) & 63) == 0) {
                voff = wo[cb + 1 + lane < ncb + 1 ? cb + 1 + lane : ncb];
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(voff));
            }
            vnext = ent[__builtin_amdgcn_readlane(voff, (cb + 1) & 63) + lane];
            __builtin_amdgcn_sched_barrier(0);
            issue_tile(cb + 1, (cb + 1) & 1);
        }
        const unsigned tbase = mg_lds_addr(smem) + lane * LB + (cb & 1) * TILEB;
        // one part of the record in `vrec`: per-row counts from the header, entries through v_readlane; two entries of a row
        // share one LDS wait.  Measured alternatives at 10 000 nodes, density 1e-2, F = 1024 (this form: 149 us): the first two
        // entries of FOUR rows behind one wait 197 us (the extra scalar branches cost more than the waits they save: a CU has one
        // scalar unit for its 16 waves and this loop is scalar-driven); entry fields split on the vector side and three
        // v_readlane per entry instead of one + three scalar operations 172 us (SGPR-write hazards).
        auto run_part = [&](unsigned rec) __attribute__((always_inline)) {
#ifdef MG_SPMM_ABLATE_FIXED
            // measurement build (results wrong on purpose; tools/dev/build_variant.py): the instruction mix of a record with TWO
            // fixed-lane slots per row -- constant v_readlane indices, no counts, no loops, five rows' reads behind one wait --
            // on whatever the first lanes of the record hold (offsets masked into the tile)
            static_assert(U % 5 == 0, "ablation written for U = 10 / 20");
#pragma unroll
            for (int u0 = 0; u0 < U; u0 += 5) {
                unsigned e[10];
                xreg x[10];
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    e[k] = __builtin_amdgcn_readlane(rec, (1 + HD + 2 * u0 + k) & 63);
                    x[k] = sb_lds_read<LB>(tbase + ((e[k] >> 16) & (TILEB - 8)));
                }
                mg_lds_wait<0>();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 10; ++k) sb_fma<LB>(acc[u0 + (k >> 1)], x[k], e[k]);
            }
            return;
#endif
            unsigned cnt[HD];
#pragma unroll
            for (int i = 0; i < HD; ++i) cnt[i] = __builtin_amdgcn_readlane(rec, 1 + i);
            int j = 1 + HD;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int n = (cnt[u >> 2] >> ((u & 3) * 8)) & 0xff;
                for (; n >= 2; n -= 2) {
                    const unsigned e0 = __builtin_amdgcn_readlane(rec, j), e1 = __builtin_amdgcn_readlane(rec, j + 1);
                    const xreg x0 = sb_lds_read<LB>(tbase + (e0 >> 16));
                    const xreg x1 = sb_lds_read<LB>(tbase + (e1 >> 16));
                    mg_lds_wait<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    sb_fma<LB>(acc[u], x0, e0);
                    sb_fma<LB>(acc[u], x1, e1);
                    j += 2;
                }
                if (n) {
                    const unsigned e0 = __builtin_amdgcn_readlane(rec, j);
                    const xreg x0 = sb_lds_read<LB>(tbase + (e0 >> 16));
                    mg_lds_wait<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    sb_fma<LB>(acc[u], x0, e0);
                    j += 1;
                }
            }
        };
        run_part(vrec);
        unsigned hdr = __builtin_amdgcn_readlane(vrec, 0);
        if (!(hdr >> 31)) {
            // rare: the record continues in further parts (more than 64 - 1 - HD entries of this wave in one column block)
            unsigned part_dw = 0;
            const unsigned rec0 = wo[cb];
            do {
                part_dw += 1 + HD + (hdr & 0x7fffffffu);
                unsigned vmore = ent[rec0 + part_dw + lane];
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(vmore));
                run_part(vmore);
                hdr = __builtin_amdgcn_readlane(vmore, 0);
            } while (!(hdr >> 31));
        }
        vrec = vnext;
    }

    // epilogue: wave w owns rows rb * 16 U + w * U + u; lane = NA features
    const int fbase = slab * FS + lane * NA;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int row = (rb * NW + wave) * U + u;
        if (row < n_rows) {
            float o[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) o[a] = mg_act(acc[u][a], act);
            if (y_bf16) {
                uint16_t* yp = static_cast<uint16_t*>(Yv) + (size_t)row * F + fbase;
                if constexpr (NA == 4) {
                    const sb_u32x2 ov = {sb_pack_bf16(o[0], o[1]), sb_pack_bf16(o[2], o[3])};
                    __builtin_nontemporal_store(ov, reinterpret_cast<sb_u32x2*>(yp));
                } else {
                    __builtin_nontemporal_store(sb_pack_bf16(o[0], o[1]), reinterpret_cast<unsigned*>(yp));
                }
            } else {
                float* yp = static_cast<float*>(Yv) + (size_t)row * F + fbase;
                if constexpr (NA == 4) {
                    __builtin_nontemporal_store(f32x4{o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4*>(yp));
                } else {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    __builtin_nontemporal_store(f32x2{o[0], o[1]}, reinterpret_cast<f32x2*>(yp));
                }
            }
        }
    }
}

// fp32 -> bf16 (round to nearest even), flat
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < n) {
        *reinterpret_cast<unsigned*>(dst + i) = sb_pack_bf16(src[i], src[i + 1]);
    } else if (i < n) {
        dst[i] = (uint16_t)(sb_pack_bf16(src[i], 0.f) & 0xffffu);
    }
}

template <int LB, int U, int BC>
int launch_tiled(const uint32_t* wave_off, const uint32_t* ent, const uint16_t* X, int n_rows, int n_cols, int F, void* Y,
                 int y_bf16, int act, hipStream_t st) {
    constexpr int FS = 32 * LB, TILEB = BC * 64 * LB;
    const int n_rb = (n_rows + 16 * U - 1) / (16 * U), ncb = (n_cols + BC - 1) / BC;
    const int total = (F / FS) * n_rb, per_xcd = (total + 7) / 8;
    auto kfn = spmm_bf16_tiled_kernel<LB, U, BC>;
    MG_DYN_LDS(kfn, 2 * TILEB);
    hipLaunchKernelGGL(kfn, dim3(8 * per_xcd), dim3(1024), 2 * TILEB, st, wave_off, ent, X, n_rows, n_cols, F, Y, y_bf16, act,
                       n_rb, ncb);
    return 0;
}

}  // namespace

// Measurement aid (tools/dev/slabcopy_exp.py): what does the ACCESS PATTERN of the slab kernels cost, with no sparse
// matrix at all?  Copies (or only reads / only writes) a [n_rows, pitch] byte matrix in pieces of `piece` bytes; piece j of
// row r belongs to XCD (j * (8 / ppr) + r % (8 / ppr)) when ppr = pitch / piece <= 8 (piece = 256 B at pitch 2 KiB is the
// slab kernels' pattern), mode bit 8 = ignore XCDs (pieces dealt round-robin to workgroups in address order).
__global__ __launch_bounds__(256) void slabcopy_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                       int n_rows, int pitch, int piece, int mode) {
    const int ppr = pitch / piece;
    const int lpp = piece / 16;                                   // lanes per piece
    const int ppw = 256 / lpp;                                    // pieces per workgroup pass
    const int t = threadIdx.x, pl = t / lpp, off = (t % lpp) * 16;
    const bool linear = mode & 256;
    const int what = mode & 3;
    const bool nt = !(mode & 16);
    u32x4 sink = {0u, 0u, 0u, 0u};
    if (linear) {
        const long long total = (long long)n_rows * ppr;
        for (long long p = (long long)blockIdx.x * ppw + pl; p < total; p += (long long)gridDim.x * ppw) {
            const size_t a = (size_t)(p / ppr) * pitch + (size_t)(p % ppr) * piece + off;
            u32x4 v = {1u, 2u, 3u, 4u};
            if (what != 2) v = *reinterpret_cast<const u32x4*>(src + a);
            if (what == 1) {
                sink[0] ^= v[0] ^ v[1] ^ v[2] ^ v[3];
            } else if (nt) {
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + a));
            } else {
                *reinterpret_cast<u32x4*>(dst + a) = v;
            }
        }
    } else {
        const int xcd = blockIdx.x & 7, bj = blockIdx.x >> 3, nb = gridDim.x >> 3;
        const int share = ppr >= 8 ? 1 : 8 / ppr;                 // XCDs sharing one piece column
        // this XCD's pieces: columns j with j * share / ... ; enumerate (row, col) pairs owned by the XCD
        const int ncol = ppr >= 8 ? ppr / 8 : 1;                  // piece columns owned per XCD
        const int col0 = ppr >= 8 ? xcd * ncol : xcd / share;
        const int rphase = ppr >= 8 ? 0 : xcd % share;
        const int rstep = share;
        const long long owned_rows = (n_rows - rphase + rstep - 1) / rstep;
        const long long total = owned_rows * ncol;
        for (long long p = (long long)bj * ppw + pl; p < total; p += (long long)nb * ppw) {
            const int r = rphase + (int)(p / ncol) * rstep, j = col0 + (int)(p % ncol);
            const size_t a = (size_t)r * pitch + (size_t)j * piece + off;
            u32x4 v = {1u, 2u, 3u, 4u};
            if (what == 3) {
                // k pseudo-random gathers of the same piece column (the re-reads of the sparse product), all in flight at once
                const int k = (mode >> 12) & 15;
                u32x4 gv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    gv[i] = u32x4{0u, 0u, 0u, 0u};
                    if (i < k) {
                        const unsigned h = ((unsigned)(p * 8 + i) * 2654435761u) >> 8;
                        gv[i] = *reinterpret_cast<const u32x4*>(src + (size_t)(h % (unsigned)n_rows) * pitch + (size_t)j * piece + off);
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) sink[0] ^= gv[i][0] ^ gv[i][1] ^ gv[i][2] ^ gv[i][3];
                continue;
            }
            if (what != 2) v = *reinterpret_cast<const u32x4*>(src + a);
            if (what == 1) {
                sink[0] ^= v[0] ^ v[1] ^ v[2] ^ v[3];
            } else if (nt) {
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + a));
            } else {
                *reinterpret_cast<u32x4*>(dst + a) = v;
            }
        }
    }
    if ((what == 1 || what == 3) && sink[0] == 0x12345u) *reinterpret_cast<u32x4*>(dst) = sink;
}

extern "C" int mgnns_debug_slabcopy(const void* src, void* dst, int n_rows, int pitch, int piece, int mode, int wgx,
                                    mgnns_stream_t stream) {
    MG_REQUIRE(src && dst && n_rows > 0 && piece >= 16 && piece <= 4096 && pitch % piece == 0 && (piece & (piece - 1)) == 0 && wgx > 0,
               "mgnns_debug_slabcopy: bad arguments");
    hipLaunchKernelGGL(slabcopy_kernel, dim3(8 * wgx), dim3(256), 0, (hipStream_t)stream, static_cast<const unsigned char*>(src),
                       static_cast<unsigned char*>(dst), n_rows, pitch, piece, mode);
    MG_CHECK_LAUNCH("mgnns_debug_slabcopy");
    return 0;
}

extern "C" int mgnns_cast_bf16(const float* src, long long n, void* dst, mgnns_stream_t stream) {
    MG_REQUIRE(src && dst && n >= 0, "mgnns_cast_bf16: null pointer or n=%lld", n);
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, (hipStream_t)stream, src,
                       static_cast<uint16_t*>(dst), n);
    MG_CHECK_LAUNCH("mgnns_cast_bf16");
    return 0;
}

extern "C" int mgnns_spmm_csr_bf16_fwd(const int32_t* row_ptr, const int32_t* col, const void* val_bf16, int n_rows, int nnz,
                                       const void* X, int F, void* Y, int y_bf16, int act, int variant,
                                       const int32_t* row_map, mgnns_stream_t stream) {
    MG_REQUIRE(row_ptr && X && Y && (nnz == 0 || (col && val_bf16)), "mgnns_spmm_csr_bf16_fwd: null pointer");
    MG_REQUIRE(n_rows >= 0 && nnz >= 0 && F > 0 && F % 8 == 0, "mgnns_spmm_csr_bf16_fwd: F=%d must be a positive multiple of 8", F);
    MG_REQUIRE(mg_aligned16(X) && mg_aligned16(Y), "mgnns_spmm_csr_bf16_fwd: X/Y must be 16-byte aligned");
    MG_REQUIRE(act >= 0 && act <= 2, "mgnns_spmm_csr_bf16_fwd: unknown activation %d", act);
    if (n_rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const uint16_t* v = static_cast<const uint16_t*>(val_bf16);
    const uint16_t* x = static_cast<const uint16_t*>(X);
    const int spx = ((F + 127) / 128 + 7) / 8;
    // variant 0: by feature width (measured cache-cold at 10 000 nodes, density 4e-4: F = 1024 ring 20.7 us / register 24.2 us,
    // F = 2048 ring 37.2 us / register with two slabs per pass 33.9 us).  Sweep codes (tools/dev):
    //   1 << 30 | (workgroups per XCD) << 4 | (NS - 1) << 2 | (RU - 1)      the register form
    //   1 << 29 | (workgroups per XCD) << 8 | (RI == 8) << 1 | (NSL - 1)     the ring form
    if (variant == 0) variant = spx >= 2 ? ((1 << 30) | (384 << 4) | 4) : ((1 << 29) | (128 << 8) | 2);
    if (variant & (1 << 30)) {
        variant &= ~(1 << 30);
        const int wgx = variant >> 4, ns = ((variant >> 2) & 3) + 1, ru = (variant & 3) + 1;
        MG_REQUIRE(wgx > 0 && wgx <= 4096 && ns <= 2 && ru <= 2, "mgnns_spmm_csr_bf16_fwd: bad variant %d", variant);
        const dim3 grid(8 * wgx), blk(256);
        if (ns == 1 && ru == 1)
            hipLaunchKernelGGL((spmm_bf16_slab_kernel<1, 1>), grid, blk, 0, st, row_ptr, col, v, x, n_rows, F, Y, y_bf16, act, row_map);
        else if (ns == 1)
            hipLaunchKernelGGL((spmm_bf16_slab_kernel<1, 2>), grid, blk, 0, st, row_ptr, col, v, x, n_rows, F, Y, y_bf16, act, row_map);
        else if (ru == 1)
            hipLaunchKernelGGL((spmm_bf16_slab_kernel<2, 1>), grid, blk, 0, st, row_ptr, col, v, x, n_rows, F, Y, y_bf16, act, row_map);
        else
            hipLaunchKernelGGL((spmm_bf16_slab_kernel<2, 2>), grid, blk, 0, st, row_ptr, col, v, x, n_rows, F, Y, y_bf16, act, row_map);
    } else {
        MG_REQUIRE(variant & (1 << 29), "mgnns_spmm_csr_bf16_fwd: bad variant %d", variant);
        variant &= ~(1 << 29);
        const int wgx = variant >> 8, ri8 = (variant >> 1) & 1, nsl = (variant & 1) + 1;
        MG_REQUIRE(wgx > 0 && wgx <= 4096, "mgnns_spmm_csr_bf16_fwd: bad variant %d", variant);
        const dim3 grid(8 * wgx), blk(256);
#define MG_RING(NSL_, RI_)                                                                                       \
    if (nsl == NSL_ && (ri8 ? 8 : 16) == RI_) {                                                                  \
        auto kfn = spmm_bf16_ring_kernel<NSL_, RI_>;                                                              \
        const int lds = 4 * (RI_ * 1024 + 1536);                                                                  \
        MG_DYN_LDS(kfn, lds);                                                                                     \
        hipLaunchKernelGGL(kfn, grid, blk, lds, st, row_ptr, col, v, x, n_rows, F, Y, y_bf16, act, nnz, row_map); \
    }
        MG_RING(1, 16) MG_RING(1, 8) MG_RING(2, 16) MG_RING(2, 8)
#undef MG_RING
    }
    MG_CHECK_LAUNCH("mgnns_spmm_csr_bf16_fwd");
    return 0;
}

extern "C" int mgnns_spmm_tiled_bf16_fwd(const uint32_t* wave_off, const uint32_t* ent, int lane_bytes, int rows_per_wave,
                                         int tile_cols, int n_rows, int n_cols, const void* X, int F, void* Y, int y_bf16,
                                         int act, mgnns_stream_t stream) {
    MG_REQUIRE(wave_off && ent && X && Y, "mgnns_spmm_tiled_bf16_fwd: null pointer");
    MG_REQUIRE(n_rows > 0 && n_cols > 0, "mgnns_spmm_tiled_bf16_fwd: n_rows=%d n_cols=%d", n_rows, n_cols);
    MG_REQUIRE(act >= 0 && act <= 2, "mgnns_spmm_tiled_bf16_fwd: unknown activation %d", act);
    MG_REQUIRE(mg_aligned16(X) && mg_aligned16(Y), "mgnns_spmm_tiled_bf16_fwd: X/Y must be 16-byte aligned");
    MG_REQUIRE(lane_bytes == 8 || lane_bytes == 4, "mgnns_spmm_tiled_bf16_fwd: lane_bytes=%d (4 or 8)", lane_bytes);
    MG_REQUIRE(F > 0 && F % (32 * lane_bytes) == 0, "mgnns_spmm_tiled_bf16_fwd: F=%d must be a multiple of %d", F, 32 * lane_bytes);
    hipStream_t st = (hipStream_t)stream;
    const uint16_t* x = static_cast<const uint16_t*>(X);
    int rc = -1;
#define MG_TILED(LB, U, BC) \
    if (lane_bytes == LB && rows_per_wave == U && tile_cols == BC) rc = launch_tiled<LB, U, BC>(wave_off, ent, x, n_rows, n_cols, F, Y, y_bf16, act, st)
    MG_TILED(8, 10, 128);
    MG_TILED(8, 20, 128);
    MG_TILED(4, 20, 256);
    MG_TILED(4, 10, 256);
#undef MG_TILED
    MG_REQUIRE(rc != -1, "mgnns_spmm_tiled_bf16_fwd: no kernel for lane_bytes=%d rows_per_wave=%d tile_cols=%d", lane_bytes,
               rows_per_wave, tile_cols);
    if (rc) return rc;
    MG_CHECK_LAUNCH("mgnns_spmm_tiled_bf16_fwd");
    return 0;
}
