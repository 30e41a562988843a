// Declarations shared between the translation units of librfn_hip.so (not part of the C ABI): the prepared form of the per-step
// launches of the stage-II / decoder recurrences, which the persistent recurrence kernels (rfn_chain.hip) run inside one launch.
#pragma once
#include "rfn_attn_small_body.h"
#include "rfn_cellgemm_body.h"
#include "rfn_deccell_body.h"

struct CgPrepared {      // rfn_cell_gemm's launch, not launched (rfn_cellgemm.hip)
    CgArgs a;
    int variant;         // tile variant 1 / 2 / 3
    int blocks;          // tiles of the launch
    int epi;             // CG_EPI_*
    bool bkf;            // B operands are [n][k] (forward products)
    bool deep;           // RFN_CELL_VARIANT_DEEP: take the deep-ring kernel when the tiles do not outnumber the CUs
};
int rfn_cg_prepare(int M, int nout, const rfn_cell_out* outs, int R, float drop_p, uint64_t seed, int variant, CgPrepared* pz);
int rfn_cg_launch(const CgPrepared& pz, void* stream);
int rfn_cg_replan32(CgPrepared* pz);      // re-tile a prepared launch on the 32-row variant (same results)
// The decoder's hoisted attention backward (rfn_deccell.hip): argument block without the launch, whether its rows take the fast
// body, and a prepared store-epilogue dX product launched WITH those rows in one grid (rfn_cellgemm.hip; RFN_ERR_UNSUPPORTED =
// nothing launched, the caller issues the two launches).
int rfn_dec_attn_bwd_args(const float* proj, int64_t psb, int64_t psl, const float* hproj, const float* w_out, const float* alpha,
                          const float* U, int64_t usb, int64_t usl, const float* dgates, int64_t ldg, int B, int L, int A, int GD,
                          float* dproj, int64_t dpsb, int64_t dpsl, int accumulate, float* dhproj, float* dw_part, DecAttnBwdArgs* out);
bool rfn_dec_attn_bwd_fast_ok(const DecAttnBwdArgs& a);
int rfn_cg_launch_with_rows(const CgPrepared& pz, const DecAttnBwdArgs& rows_args, int rows, void* stream);

struct AttnSmallPrepared {   // rfn_attn_small_fwd / _bwd's launch, not launched (rfn_attn.hip)
    AttnSmallArgs a;
    int B, ngroups;
    int backward, vec;
    size_t lds;
};
int rfn_attn_small_prepare_fwd(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl, const float* const* hproj,
                               const float* const* w_out, const float* const* b_out, const float* const* att_seq, int64_t sb,
                               int64_t sl, int B, int L, int A, int D, float* const* alpha, float* const* z, int64_t ldz,
                               AttnSmallPrepared* pz);
int rfn_attn_small_prepare_bwd(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl, const float* const* hproj,
                               const float* const* w_out, const float* const* alpha, const float* const* att_seq, int64_t sb,
                               int64_t sl, const float* const* dz, int64_t lddz, int B, int L, int A, int D, float* const* dproj,
                               int64_t dproj_sb, int64_t dproj_sl, int accumulate_dproj, float* const* dhproj,
                               float* const* dw_part, float* const* datt_seq, AttnSmallPrepared* pz);
int rfn_attn_small_launch(const AttnSmallPrepared& pz, void* stream);

// One step of a recurrence chain as the three launches of rounds 3-4 (forward: K1, attention, K3 + LSTM; backward: Kb1, attention
// backward, Kb2 + LSTM backward of the step below), prepared.
struct ChainStep {
    CgPrepared g0;
    AttnSmallPrepared at;
    CgPrepared g2;
};
// Runs `nsteps` prepared steps: inside ONE persistent launch when `persist` is set and the chain qualifies (rfn_chain.hip says
// what that takes), as 3 * nsteps ordinary launches otherwise.  `bar`: >= RFN_CHAIN_BAR_WORDS zeroable 32-bit words of the
// caller's workspace (the grid barrier's counters).  Results are bit-identical either way.
#define RFN_CHAIN_BAR_WORDS 2048
int rfn_chain_run(const ChainStep* steps, int nsteps, int persist, uint32_t* bar, void* stream);
