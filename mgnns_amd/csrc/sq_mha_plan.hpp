// Packing plan of a text mask for the packed masked attention launches (sq_mha32_bf16.hip: sq_mha32_packed_kernel): one
// workgroup's worth of device code, shared by the stand-alone plan launch (mgnns_sq_mha32_plan) and by the BiLSTM's prep launch
// (lstm.hip: an extra workgroup of lstm_prep_kernel builds the plan of the batch's mask while the others pack the batch, so
// neither masked stack of the forward pays a launch for it -- Multi_GCN_Multihead_att.py:509-527 runs both on the same mask).
#pragma once
#include "common.hpp"

namespace mg_plan {

constexpr int PR = 128;                 // rows a group holds (4 tiles of 32)
constexpr int PS = 16;                  // samples a group holds
constexpr int PLAN_HDR = 4;
constexpr int MAX_B = 4096;             // one workgroup scans the batch
__host__ __device__ constexpr size_t lds_bytes(int B) { return ((size_t)8 * B + 4) * sizeof(int); }
// plan[2]: the KIND of the plan (its ALIGN / ROWS / SAMP).  The two forms have the same size for a batch, so the size cannot tell
// them apart: every consumer compares this word (and plan[1] with its B) before it trusts a group's extents, and raises
// MGNNS_STATUS_BAD_PLAN instead of running on a plan of the other kind.
__host__ __device__ constexpr int kind_word(int align, int rows, int samp) { return align | (rows << 8) | (samp << 20); }

// plan = int32 [PLAN_HDR + 4 B + 2 B]: [0] number of groups, [1] B, [2] kind_word(ALIGN, ROWS, SAMP), [3] 0; group g at PLAN_HDR + 4 g: first sample, samples, rows;
// sample b at PLAN_HDR + 4 B + 2 b: first row inside its group, live rows (last unmasked position + 1).
// Greedy first fit in batch order (a group closes at 16 samples or when the next sample's 8-aligned rows would pass 128),
// computed without a serial pass over the samples: every sample finds where a group STARTING at it would end (<= 16 steps,
// all samples at once), one thread follows that chain from sample 0 (one hop per group), every group lays out its samples.
// s_plan: lds_bytes(B) of LDS -- [B] live rows, [B] row offsets, [B] next group start, [B] rows of a group from here, [4 B] groups.
// Called by ALL NT threads of one workgroup.
// ALIGN / ROWS / SAMP: a sample's rows are padded to a multiple of ALIGN, a group holds at most ROWS rows and SAMP samples.  The packed
// bf16 kernel: 8 / 128 / 16 (tiles of 32 rows, eight-row blocks); the grouped split-bf16 core: 16 / 112 / 7 (whole tiles of 16).
template <int NT, int ALIGN = 8, int ROWS = PR, int SAMP = PS>
__device__ __forceinline__ void build(const float* __restrict__ mask, int B, int L, int* __restrict__ plan, int* s_plan) {

    int* s_lv = s_plan;
    int* s_off = s_plan + B;
    int* s_next = s_plan + 2 * B;
    int* s_rows = s_plan + 3 * B;
    int* s_grp = s_plan + 4 * B;
    int* s_ng = s_plan + 8 * B;
    const int tid = threadIdx.x;
    // live rows = last unmasked position + 1.  A WAVE per sample row, four rows per trip: the row as coalesced dword loads (lane,
    // lane + 64, ...: eight loads in flight per lane for L <= 128), the last live position of a lane's elements, a 64-lane max.
    // (Round 4 swept the mask with an LDS atomicMax per live position; one thread per row with 16-byte loads was no better -- 25
    // strided loads per thread that the compiler does not keep in flight together: ~10 us either way, and with the plan riding on the
    // BiLSTM's prep launch that time sits on the chain the forward follows.)
    {
        const int lane = tid & 63, wv = tid >> 6, NW = NT / 64;
        for (int b0 = wv * 4; b0 < B; b0 += NW * 4) {
            int lv[4] = {0, 0, 0, 0};
            for (int p0 = 0; p0 < L; p0 += 128) {
                float m[4][2];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int p = p0 + 64 * h + lane;
                        m[r][h] = (b0 + r < B && p < L) ? mask[(size_t)(b0 + r) * L + p] : 0.0f;
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (m[r][0] != 0.0f) lv[r] = p0 + lane + 1;
                    if (m[r][1] != 0.0f) lv[r] = p0 + 64 + lane + 1;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int v = lv[r];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const int u = __shfl_xor(v, o, 64);
                    v = u > v ? u : v;
                }
                if (lane == 0 && b0 + r < B) s_lv[b0 + r] = v;
            }
        }
    }
    __syncthreads();
    for (int b = tid; b < B; b += NT) {
        int rows = 0, j = b;
        while (j < B && j - b < SAMP) {
            const int lv = s_lv[j];
            const int l8 = lv <= ALIGN ? ALIGN : (lv + ALIGN - 1) & ~(ALIGN - 1);
            if (rows + l8 > ROWS) break;
            rows += l8;
            ++j;
        }
        s_next[b] = j;
        s_rows[b] = rows;
    }
    __syncthreads();
    if (tid == 0) {
        int g = 0;
        for (int b = 0; b < B; ++g) {                    // one hop per group
            s_grp[4 * g] = b;
            b = s_next[b];
        }
        *s_ng = g;
        plan[0] = g;
        plan[1] = B;
        plan[2] = kind_word(ALIGN, ROWS, SAMP);
        plan[3] = 0;
    }
    __syncthreads();
    const int ng = *s_ng;
    for (int g = tid; g < ng; g += NT) {
        const int b0 = s_grp[4 * g];
        s_grp[4 * g + 1] = s_next[b0] - b0;
        s_grp[4 * g + 2] = s_rows[b0];
        s_grp[4 * g + 3] = 0;
        int rows = 0;
        for (int b = b0; b < s_next[b0]; ++b) {
            const int lv = s_lv[b];
            s_off[b] = rows;
            rows += lv <= ALIGN ? ALIGN : (lv + ALIGN - 1) & ~(ALIGN - 1);
        }
    }
    __syncthreads();
    for (int i = tid; i < 4 * ng; i += NT) plan[PLAN_HDR + i] = s_grp[i];
    for (int b = tid; b < B; b += NT) {
        plan[PLAN_HDR + 4 * B + 2 * b] = s_off[b];
        plan[PLAN_HDR + 4 * B + 2 * b + 1] = s_lv[b];
    }
}

}  // namespace mg_plan
