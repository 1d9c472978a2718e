// The verify-loop body of MagicDec's speculative decoding as ONE integer kernel
// (row a1 of SURVEY.md section 8): accept mask, accept_nums, length rollback /
// advance for the target and draft page tables, output scatter, bonus token,
// termination test and the "double buffer" for rows that accepted all gamma.
//
// reference: tests/SnapKV/longspec_benchmark.py:208-285 (StreamingLLM twin
// :205-282), tests/SnapKV/selfspec_benchmark.py:145-211,
// tests/StreamingLLM/selfspec_benchmark.py:159-238.
// The reference runs ~15 tiny ATen kernels and 4 host syncs per iteration; here
// the host reads one flag word.  All arithmetic is integer => bit-exact.
#include "md_common.h"

namespace {

struct AcceptParams {
    int64_t* tokens_buffer;        // [B, gamma+1]
    const int64_t* target_tokens;  // [B, gamma+1]
    int64_t* output;               // [B, out_cols]
    int64_t* num_nodes;            // [B]
    int32_t* cachelens;            // [B] target, already += gamma+1
    int32_t* last_page_len;        // [B]
    int32_t* draft_cachelens;      // [B] or null
    int32_t* draft_last_page_len;  // [B] or null
    int64_t* accept_nums;          // [B]
    int64_t* bonus;                // [B]
    int64_t* double_buffer;        // [B,2] or null
    int64_t* cachelens_update;     // [B] or null
    int32_t* flags;                // [0]=terminal [1]=next_double
    int64_t eot_1, eot_2, max_nodes;
    int B, gamma, out_cols, draft_rollback, draft_cap;
};

__global__ __launch_bounds__(1024) void accept_kernel(const AcceptParams p) {
    __shared__ int s_term, s_full;
    if (threadIdx.x == 0) {
        s_term = 0;
        s_full = 0;
    }
    __syncthreads();
    const int G = p.gamma;
    // pass 1: per-row accept count, rollback, scatter, bonus
    for (int b = threadIdx.x; b < p.B; b += blockDim.x) {
        const int64_t* tb = p.tokens_buffer + (int64_t)b * (G + 1);
        const int64_t* tt = p.target_tokens + (int64_t)b * (G + 1);
        int acc = 0;
        bool alive = true;
        for (int j = 0; j < G; ++j) {
            const int64_t d = tb[j + 1];
            const bool eot = (d == p.eot_1) || (d == p.eot_2);
            alive = alive && (tt[j] == d) && !eot;  // cumprod(flag & ~eot)
            acc += alive ? 1 : 0;
            // reference's `condition = (eot & accept).any()` can never fire: accept already excludes eot
        }
        const int an = acc + 1;
        p.accept_nums[b] = an;
        // rollback the target by gamma+1, scatter accepted tokens, advance by accept_nums
        const int base = p.cachelens[b] - (G + 1);
        for (int j = 0; j < an; ++j) {
            const int col = base + j;
            if (col >= 0 && col < p.out_cols) p.output[(int64_t)b * p.out_cols + col] = tb[j];
        }
        p.cachelens[b] = base + an;
        p.last_page_len[b] = p.last_page_len[b] - (G + 1) + an;
        if (p.draft_cachelens) {
            const int adv = an < p.draft_cap ? an : p.draft_cap;
            p.draft_cachelens[b] = p.draft_cachelens[b] - p.draft_rollback + adv;
            p.draft_last_page_len[b] = p.draft_last_page_len[b] - p.draft_rollback + adv;
        }
        const int64_t bon = tt[an - 1];
        p.bonus[b] = bon;
        const int64_t nn = p.num_nodes[b] + an;
        p.num_nodes[b] = nn;
        if (bon == p.eot_1 || bon == p.eot_2 || nn >= p.max_nodes) atomicOr(&s_term, 1);
        if (an == G + 1) atomicOr(&s_full, 1);
    }
    __syncthreads();
    const int term = s_term, full = s_full;
    // pass 2: prepare the next iteration (or finish the batch)
    for (int b = threadIdx.x; b < p.B; b += blockDim.x) {
        int64_t* tb = p.tokens_buffer + (int64_t)b * (G + 1);
        const int64_t bon = p.bonus[b];
        if (!term) {
            tb[0] = bon;
            if (full && p.double_buffer) {
                const bool m = p.accept_nums[b] == G + 1;
                p.double_buffer[b * 2 + 0] = m ? tb[G] : bon;
                p.double_buffer[b * 2 + 1] = m ? bon : 0;
                p.cachelens_update[b] = m ? 2 : 1;
            }
        } else {
            const int64_t nn = p.num_nodes[b];
            if (nn >= 0 && nn < p.out_cols) p.output[(int64_t)b * p.out_cols + nn] = bon;
            p.num_nodes[b] = nn + 1;
        }
    }
    if (threadIdx.x == 0) {
        p.flags[0] = term;
        p.flags[1] = (!term && full && p.double_buffer) ? 1 : 0;
    }
}

}  // namespace

extern "C" int md_accept_rollback(int64_t* tokens_buffer, const int64_t* target_tokens, int64_t* output,
                                  int out_cols, int64_t* num_nodes, int32_t* cachelens, int32_t* last_page_len,
                                  int32_t* draft_cachelens, int32_t* draft_last_page_len, int B, int gamma,
                                  int draft_rollback, int draft_cap, int64_t eot_1, int64_t eot_2,
                                  int64_t max_nodes, int64_t* accept_nums, int64_t* bonus, int64_t* double_buffer,
                                  int64_t* cachelens_update, int32_t* flags, md_stream_t stream) {
    MD_CHECK_ARG(tokens_buffer && target_tokens && output && num_nodes && cachelens && last_page_len && accept_nums &&
                     bonus && flags,
                 "md_accept_rollback: null pointer argument");
    MD_CHECK_ARG((draft_cachelens == nullptr) == (draft_last_page_len == nullptr),
                 "md_accept_rollback: draft_cachelens and draft_last_page_len go together");
    MD_CHECK_ARG((double_buffer == nullptr) == (cachelens_update == nullptr),
                 "md_accept_rollback: double_buffer and cachelens_update go together");
    MD_CHECK_ARG(B > 0 && gamma >= 1 && out_cols > 0, "md_accept_rollback: bad shape B=%d gamma=%d", B, gamma);
    AcceptParams p;
    p.tokens_buffer = tokens_buffer;
    p.target_tokens = target_tokens;
    p.output = output;
    p.num_nodes = num_nodes;
    p.cachelens = cachelens;
    p.last_page_len = last_page_len;
    p.draft_cachelens = draft_cachelens;
    p.draft_last_page_len = draft_last_page_len;
    p.accept_nums = accept_nums;
    p.bonus = bonus;
    p.double_buffer = double_buffer;
    p.cachelens_update = cachelens_update;
    p.flags = flags;
    p.eot_1 = eot_1;
    p.eot_2 = eot_2;
    p.max_nodes = max_nodes;
    p.B = B;
    p.gamma = gamma;
    p.out_cols = out_cols;
    p.draft_rollback = draft_rollback;
    p.draft_cap = draft_cap;
    const int threads = B >= 1024 ? 1024 : ((B + 63) / 64) * 64;
    hipLaunchKernelGGL(accept_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, p);
    MD_CHECK_LAUNCH("md_accept_rollback");
    return MD_OK;
}
