// nig_mixed.hip -- BASELINE config "all envs mixed-batch": ONE fused-rollout launch over a padded SoA batch
// whose contiguous 256-aligned lane segments run different environment types.
//
// A block looks its segment up in a small table passed with the kernel arguments (blocks are numbered in
// LAUNCH order: the segments with the most expensive env step first, so the cheap envs fill the tail of the
// launch), then runs that env's rollout body -- the same device function the per-env kernels wrap, hence the
// same results bit for bit.  One launch instead of one per segment: no per-kernel tails (a 150 k-lane PowerGrid
// segment alone fills the chip 1.14 times: its second round ran at 14 % occupancy), no launch fan-out.
// The price of one kernel for all envs: every block runs under ONE register allocation.  It is set for three waves
// per SIMD (168 registers; LDS, 45 KB per block, allows three).  PowerGrid's register-resident body wants ~185 and
// spilled ~70 dwords here (round 2); since round 3 its whole blocks run the LDS-resident body (nig_pg_lds.hpp, ~110).
// Round 4: ChemicalReactor and the four plants run their PAIRED bodies when the segment starts on an odd launch counter.
#include "nig_kernels.hpp"

namespace nig {

template <int OUT>
struct MixedLds {
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int BYTES =
        cmax(cmax(cmax(RolloutLds<ChemicalReactor, OUT>::BYTES, cmax(RolloutLds<PowerGrid, OUT>::BYTES, PgLds<BLOCK>::BYTES)),
                  cmax(RolloutLds<RobotAssembly, OUT>::BYTES, RolloutLds<AdvancedChemicalReactor, OUT>::BYTES)),
             cmax(cmax(RolloutLds<AdvancedPowerGrid, OUT>::BYTES, RolloutLds<HVACControl, OUT>::BYTES),
                  cmax(cmax(RolloutLds<WaterTreatment, OUT>::BYTES, RolloutLds<SteelAnnealing, OUT>::BYTES),
                       RolloutLds<SupplyChain, OUT>::BYTES)));
};

// OUT: 0 = no per-step outputs, 1 = reward + flag word rows (the padded [n_steps][ld] matrices), 2 = + the observation
// every env.step returns, as rows of a padded [n_steps][S_max][ld] trajectory (segment k fills rows < S_k of its own
// columns; rows >= S_k are never touched) -- what get_dataset stores per step (chemical_reactor.py:395-412).
template <int OUT>
__global__ void __launch_bounds__(BLOCK, 3) mixed_rollout_kernel(const MixedArgs m)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[MixedLds<OUT>::BYTES];
    // Segment of this block (block-uniform).  The table is walked with COMPILE-TIME indices and scalar selects: a
    // run-time index into the by-value argument struct makes hipcc copy all of it to scratch and turns every
    // global access of the bodies into a flat one (pointer provenance lost).  ~60 dwords x 11 selects per block,
    // once, against a quarter of a million instructions of rollout.
    RolloutArgs q = m.seg[0];
    int env = m.env[0];
    uint32_t blk0 = 0u;
#pragma unroll
    for (int j = 1; j < MIXED_MAX_SEG; ++j) {
        if (j < m.n_seg && blockIdx.x >= m.blk_end[j - 1]) { q = m.seg[j]; env = m.env[j]; blk0 = m.blk_end[j - 1]; }
    }
    const uint32_t base = (blockIdx.x - blk0) * BLOCK;         // first lane of the block inside its segment
    // A WHOLE block of a handle on which no lane can be frozen runs the unpredicated body without freeze handling (FULL,
    // NOFREEZE: counted vmcnt waits instead of a drain per step, no second copy of the state -- what the stand-alone
    // kernels' whole blocks run); a segment's ragged last block, or a handle with frozen lanes, the predicated one.
    const bool lean = base + BLOCK <= q.s.B && (q.s.hflags & NIG_F_AUTORESET) != 0 && (q.s.hflags & HF_MAY_HOLD_DONE) == 0;
    // The envs whose steps 2k-1, 2k share one generator block (ChemicalReactor, the four plants): whole blocks of a
    // segment whose FIRST step has an odd launch counter run the paired body -- one Philox block per two steps, the
    // normals produced a step ahead (rollout_body PAIRED; the stand-alone launches peel a misaligned first step on the
    // host, here a segment that starts on an even counter simply runs the unpaired body: same values either way).
    const uint32_t t_first = (q.s.t_ptr ? *q.s.t_ptr : 0u) + q.s.t_off + (uint32_t)q.it0 + 1u;     // block-uniform
    const bool paired = lean && (t_first & 1u) != 0u;
#define NIG_MIXED_BODY(Env)                                                   \
    if (lean) rollout_body<Env, OUT, false, true, true>(q, base, smem);        \
    else rollout_body<Env, OUT, false, false>(q, base, smem);                  \
    break
#define NIG_MIXED_BODY_SHARED(Env)                                            \
    if (paired) rollout_body<Env, OUT, true, true, true>(q, base, smem);       \
    else if (lean) rollout_body<Env, OUT, false, true, true>(q, base, smem);   \
    else rollout_body<Env, OUT, false, false>(q, base, smem);                  \
    break
    switch (env) {
    case NIG_ENV_CHEMICAL_REACTOR: NIG_MIXED_BODY_SHARED(ChemicalReactor);
    case NIG_ENV_POWER_GRID:
        // whole blocks: the LDS-resident body (nig_pg_lds.hpp; ~110 registers, where the register-resident one spilled ~70
        // dwords under this kernel's 168) -- bit-identical, tests/test_gpu_mixed.py
        if (lean) pg_lds_rollout_body<OUT, BLOCK>(q, base, smem);
        else rollout_body<PowerGrid, OUT, false, false>(q, base, smem);
        break;
    case NIG_ENV_ROBOT_ASSEMBLY: NIG_MIXED_BODY(RobotAssembly);
    case NIG_ENV_ADV_CHEMICAL_REACTOR: NIG_MIXED_BODY(AdvancedChemicalReactor);
    case NIG_ENV_ADV_POWER_GRID: NIG_MIXED_BODY(AdvancedPowerGrid);
    case NIG_ENV_HVAC_CONTROL: NIG_MIXED_BODY_SHARED(HVACControl);
    case NIG_ENV_WATER_TREATMENT: NIG_MIXED_BODY_SHARED(WaterTreatment);
    case NIG_ENV_STEEL_ANNEALING: NIG_MIXED_BODY_SHARED(SteelAnnealing);
    default: NIG_MIXED_BODY_SHARED(SupplyChain);
    }
#undef NIG_MIXED_BODY
#undef NIG_MIXED_BODY_SHARED
}

}  // namespace nig

void nig_launch_mixed_rollout(int out_mode, const nig::MixedArgs &m, unsigned grid, hipStream_t st)
{
    using namespace nig;
    if (out_mode == 0) hipLaunchKernelGGL((mixed_rollout_kernel<0>), dim3(grid), dim3(BLOCK), 0, st, m);
    else if (out_mode == 1) hipLaunchKernelGGL((mixed_rollout_kernel<1>), dim3(grid), dim3(BLOCK), 0, st, m);
    else hipLaunchKernelGGL((mixed_rollout_kernel<2>), dim3(grid), dim3(BLOCK), 0, st, m);
}
