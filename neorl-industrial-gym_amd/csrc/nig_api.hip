// nig_api.hip -- C ABI of libnig.so (include/nig.h) and the environment-independent kernels.
// The per-environment kernels are instantiated in env_*.hip and reached through nig::EnvLaunch.
#include <dlfcn.h>

#include <atomic>

#include "nig_kernels.hpp"

namespace nig {

__global__ void __launch_bounds__(BLOCK) init_ws_kernel(uint32_t *ctr, long long *life, double *ep_ret, double *tally,
                                                        int64_t ld, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= ld) return;
    ctr[i] = (i < B) ? NIG_CTR_DONE : NIG_CTR_DONE;   // nothing steps before the first reset
    life[i] = 0;
    if (ep_ret) ep_ret[i] = 0.0;
    if (tally) {
#pragma unroll
        for (int r = 0; r < NIG_T_ROWS; ++r) tally[(int64_t)r * ld + i] = 0.0;
        tally[(int64_t)NIG_T_RET_MIN * ld + i] = __builtin_inf();
        tally[(int64_t)NIG_T_RET_MAX * ld + i] = -__builtin_inf();
    }
}

__global__ void set_u32_kernel(uint32_t *p, uint32_t v) { *p = v; }

// row-major action ring [R][B][A] (slot stride `aos_stride` floats) -> rows [R][A][ld]: one thread per lane, slots blockIdx.y,
// blockIdx.y + gridDim.y, ... (the grid's y extent is capped at 65 535)
__global__ void __launch_bounds__(BLOCK) action_rows_kernel(const float *aos, int64_t aos_stride, float *soa, int64_t ld, int64_t B, int A, int slots)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= B) return;
    for (int r = (int)blockIdx.y; r < slots; r += (int)gridDim.y) {
        const float *src = aos + (int64_t)r * aos_stride + i * A;
        float *dst = soa + (int64_t)r * A * ld + i;
        for (int k = 0; k < A; ++k) dst[(int64_t)k * ld] = src[k];
    }
}

// nig_clock_stamp: one wave per block, many more blocks than compute units; a block stamps the slot of the compute unit it
// runs on -- slot = XCD (HW_REG_XCC_ID) x 256 + HW_REG_HW_ID's {se_id, sh_id, cu_id} -- because s_memtime is NOT one counter per
// chip (round 5, first attempt: per-XCD slots written by whichever block came last gave 2 030 .. 5 570 "MHz" across the XCDs of
// one run).  Waves of one compute unit write the same pair of words with values a few cycles apart (any of them will do).
__global__ void __launch_bounds__(64) clock_stamp_kernel(unsigned long long *o)
{
    const unsigned long long t = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;        // HW_REG_XCC_ID (id 20), bits 3:0
    const unsigned cu = __builtin_amdgcn_s_getreg((8 - 1) << 11 | 8 << 6 | 4) & 0xFFu;       // HW_REG_HW_ID (id 4), bits 15:8: cu_id, sh_id, se_id
    const unsigned slot = xcc * 256u + cu;
    if (threadIdx.x == 0) { o[2 * slot] = t; o[2 * slot + 1] = r; }
}

// n_total = the handle's enabled built-in constraints (3, 4 for AdvancedChemicalReactor, fewer after
// nig_set_constraint_mask): SafetyMetrics.total_constraints, base.py:115
__global__ void __launch_bounds__(BLOCK) safety_metrics_kernel(const uint32_t *flags, int32_t *out, int64_t ld_out,
                                                               int64_t B, int n_total)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= B) return;
    const uint32_t f = flags[i];
    const int nv = (int)((f >> NIG_FLAG_NVIOL_SHIFT) & 3u) + ((f & NIG_FLAG_NVIOL_HI) ? 4 : 0);
    const int nc = (int)((f >> NIG_FLAG_NCRIT_SHIFT) & 3u);
    out[0 * ld_out + i] = n_total - nv;   // constraints_satisfied   base.py:96-106
    out[1 * ld_out + i] = n_total;        // total_constraints       base.py:115
    out[2 * ld_out + i] = nv;             // violation_count
    out[3 * ld_out + i] = nc;             // critical_violations
    out[4 * ld_out + i] = n_total - nv;   // safety_score * total    base.py:116
}

// deterministic two-stage reduction of the tally rows: fixed grid, fixed tree order
__global__ void __launch_bounds__(BLOCK) reduce_tally_stage1(const double *tally, int64_t ld, int64_t B, double *scratch)
{
    __shared__ double sh[BLOCK];
    for (int r = 0; r < NIG_T_ROWS; ++r) {
        const bool is_min = (r == NIG_T_RET_MIN), is_max = (r == NIG_T_RET_MAX);
        double acc = is_min ? __builtin_inf() : (is_max ? -__builtin_inf() : 0.0);
        for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < B; i += (int64_t)gridDim.x * BLOCK) {
            const double v = tally[(int64_t)r * ld + i];
            acc = is_min ? fmin(acc, v) : (is_max ? fmax(acc, v) : acc + v);
        }
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int w = BLOCK / 2; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) {
                const double o = sh[threadIdx.x + w];
                sh[threadIdx.x] = is_min ? fmin(sh[threadIdx.x], o) : (is_max ? fmax(sh[threadIdx.x], o) : sh[threadIdx.x] + o);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) scratch[(int64_t)blockIdx.x * NIG_T_ROWS + r] = sh[0];
        __syncthreads();
    }
}

// Second stage: the block partials of stage 1 in block order, one thread per tally row.  The sums keep their fixed
// sequential order (bit-reproducible, and the same bits as rounds 1-2), but the partials are brought into LDS by the
// whole block first: read one by one from global memory by 13 threads, the 256 dependent load -> add steps took 65 us
// on the critical path of the path's one cross-GPU exchange; from LDS they take ~1 us.
__global__ void __launch_bounds__(REDUCE_BLOCKS) reduce_tally_stage2(const double *scratch, int nblk, double *out)
{
    __shared__ double sh[REDUCE_BLOCKS * NIG_T_ROWS];
    for (int i = (int)threadIdx.x; i < nblk * NIG_T_ROWS; i += REDUCE_BLOCKS) sh[i] = scratch[i];
    __syncthreads();
    const int r = threadIdx.x;
    if (r >= NIG_T_ROWS) return;
    const bool is_min = (r == NIG_T_RET_MIN), is_max = (r == NIG_T_RET_MAX);
    double acc = is_min ? __builtin_inf() : (is_max ? -__builtin_inf() : 0.0);
    for (int b = 0; b < nblk; ++b) {
        const double v = sh[b * NIG_T_ROWS + r];
        acc = is_min ? fmin(acc, v) : (is_max ? fmax(acc, v) : acc + v);
    }
    out[r] = acc;
}

// float64 action rows -> float32 (envs whose own arithmetic converts the action on entry)
__global__ void __launch_bounds__(BLOCK) narrow_rows_kernel(const double *src, int64_t ld_src, float *dst, int64_t ld_dst,
                                                            int rows, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= B) return;
    for (int k = 0; k < rows; ++k) dst[(int64_t)k * ld_dst + i] = (float)src[(int64_t)k * ld_src + i];
}

// combine `n` partial tally vectors [n][NIG_T_ROWS] in index order: sums, min / max rows (one thread per row)
__global__ void combine_partials_kernel(const double *parts, int n, double *out)
{
    const int r = threadIdx.x;
    if (r >= NIG_T_ROWS) return;
    const bool is_min = (r == NIG_T_RET_MIN), is_max = (r == NIG_T_RET_MAX);
    double acc = is_min ? __builtin_inf() : (is_max ? -__builtin_inf() : 0.0);
    for (int b = 0; b < n; ++b) {
        const double v = parts[(int64_t)b * NIG_T_ROWS + r];
        acc = is_min ? fmin(acc, v) : (is_max ? fmax(acc, v) : acc + v);
    }
    out[r] = acc;
}

__global__ void __launch_bounds__(BLOCK) copy_rows_kernel(const float *src, int64_t ld_src, float *dst, int64_t ld_dst,
                                                          int rows, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= B) return;
    for (int k = 0; k < rows; ++k) dst[(int64_t)k * ld_dst + i] = src[(int64_t)k * ld_src + i];
}

}  // namespace nig

// =====================================================================================
// host side: C ABI
// =====================================================================================
using namespace nig;

struct nig_handle {
    int env;
    int device;
    unsigned cus;          // compute units of `device`: default of the kernel-form thresholds (nig_tune)
    int64_t B;
    uint64_t seed, env0;
    int max_steps;
    double dt;
    uint32_t flags;
    uint32_t t;            // RNG launch counter
    uint32_t cmask;        // enabled built-in constraints (bit k)
    nig_layout lay;
    char *ws;
    bool owns_ws;
    double *scratch;       // reduce scratch [REDUCE_BLOCKS][NIG_T_ROWS] (inside workspace tail)
    uint32_t *t_dev;       // device copy of t read by graph-replayed step kernels
    float *state;          // state rows: inside the workspace, or an array bound by the caller
    int64_t ld_state;
    nig_policy *pol_dev;   // device copy of the installed policy
    nig_policy pol_host;   // staging copy (must outlive the async H2D copy)
    bool has_policy;
    float *mlp_stream;     // device copy of the MFMA operand stream of the MLP actor (owned)
    float *act32;          // nig_step64 on an env that takes float32 actions: the narrowed rows [A][ld] (owned, lazy)
    float *pid_mem;        // PID policies: per-lane integral / previous error, float [2*A][ld] (owned, lazy)
    bool may_hold_done;    // some lane may carry NIG_CTR_DONE although the handle auto-resets (see HF_MAY_HOLD_DONE)
    char *hst_pinned;      // host-buffer entry points: pinned staging + its device mirror (owned, lazy)
    char *hst_dev;
    size_t hst_bytes;
    float *mirror; uint32_t ld_mirror;    // StepArgs::mirror of the next step launch (nig_step_host*), else NULL
    float *act_soa; size_t act_soa_floats;   // nig_rollout with a row-major action ring on a kernel form that reads rows: the [A][ld] copy (owned, lazy)
};

struct nig_plan {
    nig_handle *h;
    int n_steps;
    hipGraph_t graph;
    hipGraphExec_t exec;
};

static unsigned grid_for(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *detail = "")
{
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) return fail(NIG_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

static const nig_env_spec SPECS[NIG_NUM_ENVS] = {
    {12, 3, 3, 500, 2, 8, 0.1, {-100.0, -50.0, -25.0}, {1, 1, 0}, 1},
    {32, 8, 3, 1000, 23, 31, 0.1, {-50.0, -30.0, -20.0}, {1, 1, 0}, 0},
    {24, 7, 3, 1000, 0, 7, 0.1, {-100.0, -200.0, -50.0}, {1, 1, 0}, 0},
    /* Advanced envs: 4 / 3 safety-metric conditions, no penalties through the base loop, deterministic */
    {20, 6, 4, 1000, 0, 0, 0.1, {0.0, 0.0, 0.0}, {0, 0, 0}, 0},
    {32, 8, 3, 500, 0, 0, 0.1, {0.0, 0.0, 0.0}, {0, 0, 0}, 0},
    /* build-specified plants (spec_plants.py): dims and constraint tables come from the generated data */
#define NIG_SPEC_ROW(K) {SpecPlant<K>::S, SpecPlant<K>::A, 3, SpecPlant<K>::MAX_STEPS, SpecPlant<K>::KS, SpecPlant<K>::KR, 0.1, \
                         {NIG_SPEC_PLANTS[K].pen[0], NIG_SPEC_PLANTS[K].pen[1], NIG_SPEC_PLANTS[K].pen[2]},                    \
                         {NIG_SPEC_PLANTS[K].crit[0], NIG_SPEC_PLANTS[K].crit[1], NIG_SPEC_PLANTS[K].crit[2]}, 1}
    NIG_SPEC_ROW(0), NIG_SPEC_ROW(1), NIG_SPEC_ROW(2), NIG_SPEC_ROW(3),
#undef NIG_SPEC_ROW
};
static const char *NAMES[NIG_NUM_ENVS] = {"ChemicalReactor-v0", "PowerGrid-v0", "RobotAssembly-v0",
                                          "AdvancedChemicalReactor-v0", "AdvancedPowerGrid-v0",
                                          "HVACControl-v0", "WaterTreatment-v0", "SteelAnnealing-v0", "SupplyChain-v0"};

// per-environment kernel launchers (env_*.hip)
const EnvLaunch *nig_launch_cr();
const EnvLaunch *nig_launch_pg();
const EnvLaunch *nig_launch_ra();
const EnvLaunch *nig_launch_acr();
const EnvLaunch *nig_launch_apg();
const EnvLaunch *nig_launch_hvac();
const EnvLaunch *nig_launch_water();
const EnvLaunch *nig_launch_steel();
const EnvLaunch *nig_launch_supply();

static const EnvLaunch *launch_of(int env)
{
    static const EnvLaunch *const T[NIG_NUM_ENVS] = {nig_launch_cr(), nig_launch_pg(), nig_launch_ra(), nig_launch_acr(),
                                                     nig_launch_apg(), nig_launch_hvac(), nig_launch_water(),
                                                     nig_launch_steel(), nig_launch_supply()};
    return T[env];
}

// SafetyMetrics.total_constraints of this handle: built-in constraints still enabled (base.py:115,224-228)
static int enabled_constraints(const nig_handle *h);
namespace nig { unsigned split_blocks_for(unsigned cus); unsigned wide_min_blocks_for(unsigned cus); }

namespace nig { static std::atomic<unsigned> g_diag_ring_fault{0u}; }   // NIG_RING_SPIN_LIMIT builds: nig_tune(NIG_TUNE_DIAG_RING_FAULT)

static StepArgs base_step_args(const nig_handle *h)
{
    const nig_layout &L = h->lay;
    StepArgs a;
    memset(&a, 0, sizeof a);
    a.state = h->state; a.ld_state = (uint32_t)h->ld_state; a.ctr = (uint32_t *)(h->ws + L.off_ctr);
    a.life_viol = (long long *)(h->ws + L.off_life_viol);
    a.ep_ret = L.off_ep_return >= 0 ? (double *)(h->ws + L.off_ep_return) : nullptr;
    a.tally = L.off_tally >= 0 ? (double *)(h->ws + L.off_tally) : nullptr;
    a.ld = (uint32_t)L.ld; a.B = (uint32_t)h->B;
    a.env0 = h->env0; a.seed_lo = (uint32_t)h->seed; a.seed_hi = (uint32_t)(h->seed >> 32);
    a.split_blocks = nig::split_blocks_for(h->cus); a.wide_min_blocks = nig::wide_min_blocks_for(h->cus);
    a.max_steps = h->max_steps; a.dt32 = (float)h->dt; a.dt = h->dt; a.cmask = h->cmask;
    a.hflags = h->flags | (h->may_hold_done ? HF_MAY_HOLD_DONE : 0u);
    a.n_en = enabled_constraints(h);
    a.mirror = h->mirror; a.ld_mirror = h->ld_mirror;      // (set by the host-buffer entry points around their step launch)
#ifdef NIG_RING_SPIN_LIMIT
    a.ring_err = h->t_dev + 16;                    // a spare word of the 256-byte launch-counter slot (zeroed at nig_create)
    if (nig::g_diag_ring_fault.load(std::memory_order_relaxed)) a.hflags |= HF_DIAG_RING_FAULT;
#endif
    return a;
}

// NIG_RING_SPIN_LIMIT builds (test-only, nig_ring.hpp): after a launch of a cooperating-wave kernel, wait for it and turn a
// recorded ring time-out into NIG_ERR_HIP.  The production build returns at once: no hidden synchronisation there.
static int ring_check(nig_handle *h, hipStream_t st, const char *who)
{
#ifdef NIG_RING_SPIN_LIMIT
    uint32_t code = 0;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(&code, h->t_dev + 16, 4, hipMemcpyDeviceToHost));
    if (code != 0u) {
        HIP_TRY(hipMemset(h->t_dev + 16, 0, 4));
        // the counter index means different rings per kernel family: three-wave kernels (nig_split*.hpp) have
        // {produced inputs, stepped results, released slots}; PowerGrid's paired kernels (nig_pg_lds.hpp) {produced draws, consumed}
        static const char *const ring3[3] = {"inputs (producer -> stepper)", "results (stepper -> recorder)", "slots released (consumer -> producer)"};
        static const char *const ring2[3] = {"draws (producer -> stepper)", "slots released (stepper -> producer)", "(no such counter in the paired form)"};
        const char *const *ring = (h->env == NIG_ENV_POWER_GRID) ? ring2 : ring3;
        char detail[320];
        snprintf(detail, sizeof detail, "%s: a wave waited more than %d polls for count %u of ring counter %u, '%s'; every role left its loop, "
                 "results of this launch are invalid", who, (int)(NIG_RING_SPIN_LIMIT), (code >> 16) & 0x7FFFu, code & 3u, ring[(code & 3u) % 3]);
        return fail(NIG_ERR_HIP, "ring protocol time-out -- %s", detail);
    }
#else
    (void)h; (void)st; (void)who;
#endif
    return NIG_OK;
}

static int enabled_constraints(const nig_handle *h)
{
    return __builtin_popcount(h->cmask & ((1u << SPECS[h->env].n_constraints) - 1u));
}

static void dispatch_step(const nig_handle *h, const StepArgs &a, bool parity, hipStream_t st)
{
    launch_of(h->env)->step(a, parity, grid_for(h->B), st);
}

namespace nig {
// Kernel-form thresholds (nig_tune).  An explicit setting (environment variable at load, nig_tune later) is
// process-wide and atomic; without one the threshold is a property of the HANDLE's device -- its compute-unit count,
// read at nig_create (256 on an MI355X in SPX mode) -- and travels with the launch arguments, so handles on devices
// with different CU counts (partition modes) or created from several threads never see each other's value
// (it used to be one plain global set from whichever device came first).
constexpr unsigned TUNE_UNSET = 0xffffffffu;
static unsigned tune_from_env(const char *name)
{
    const char *e = getenv(name);
    return e ? (unsigned)strtoul(e, nullptr, 10) : TUNE_UNSET;
}
static std::atomic<unsigned> g_split_override{tune_from_env("NIG_SPLIT_BLOCKS")};      // 256-lane blocks per round of the three-wave form
static std::atomic<unsigned> g_wide_override{tune_from_env("NIG_WIDE_MIN_BLOCKS")};    // smallest batch, in 512-lane blocks, of the wide form
static std::atomic<unsigned> g_last_cus{256u};                                         // CU count of the device of the latest handle (reporting only)
unsigned split_blocks_for(unsigned cus)
{
    const unsigned o = g_split_override.load(std::memory_order_relaxed);
    return o != TUNE_UNSET ? o : cus;
}
unsigned wide_min_blocks_for(unsigned cus)
{
    // default: the wide form (two 512-lane blocks per CU, four waves per SIMD) from the first batch that no longer fits ONE round of
    // the 256-lane form (three blocks per CU): a round of either takes a time proportional to its waves per SIMD (4 : 3), so up to
    // 1.5 wide blocks per CU the 256-lane form's single round wins -- 196 608 lanes x 250 steps, reward + flags: 1.23 ms against
    // 1.54 ms wide; 262 144 lanes (config 3): wide 1.57 ms against 1.67 ms (profiles/r04/pg_forms.txt).  Up to round 3: `cus`.
    const unsigned o = g_wide_override.load(std::memory_order_relaxed);
    return o != TUNE_UNSET ? o : cus + cus / 2 + 1;
}
}

extern "C" {

// 0.2.0: the fast-mode generator became "nig-philox-v2" (Philox4x32-7, float32 reset draws for ChemicalReactor / PowerGrid, round 3):
// every (seed, lane, t) trajectory differs from 0.1.0's "nig-philox-v1" -- stored seeds / datasets are tied to the generator id.
// 0.3.0, "nig-philox-v3" (round 4): as v2, except that PowerGrid's eight RESET load factors are 16-bit uniforms taken from the
// low bytes of the reset normals' words (six Philox blocks per reset instead of eight): PowerGrid trajectories differ from v2's
// from the first auto-reset on; ChemicalReactor, RobotAssembly and every other env are bit-identical to v2.
// 0.4.0 (round 4): generator unchanged for the reference's envs; the four build-specified plants moved to their model "v2"
// (spec_plants.py: fused multiply-adds, steps 2k-1 / 2k sharing one generator block) -- their trajectories differ from 0.3.0's.
// 0.5.0 (round 5): generator unchanged -- every fast-mode trajectory of 0.4.0 replays bit for bit (the generator's piece index and
// PowerGrid's clips / economic sum are computed with fewer instructions, same values).  Changed results: nig_rollout_mlp for envs
// with at most four actions (ChemicalReactor, WaterTreatment) sums the head in the order of its v_mfma_f32_4x4x1 form (last-bit
// differences in the action against 0.4.0; the oracle restates the new order).  New entry point: nig_clock_stamp.  A caller
// workspace is refused only when positively identified as host / managed / foreign-device memory.
const char *nig_version(void) { return "nig 0.5.0 (gfx950; generator nig-philox-v3)"; }
const char *nig_last_error(void) { return g_err; }

int nig_tune(int32_t key, int64_t value)
{
    if (key == NIG_TUNE_DIAG_RING_FAULT) {
#ifdef NIG_RING_SPIN_LIMIT
        nig::g_diag_ring_fault.store(value != 0 ? 1u : 0u, std::memory_order_relaxed);
        return NIG_OK;
#else
        return fail(NIG_ERR_UNSUPPORTED, "nig_tune: NIG_TUNE_DIAG_RING_FAULT exists in NIG_RING_SPIN_LIMIT test builds only%s");
#endif
    }
    if ((key != NIG_TUNE_SPLIT_BLOCKS && key != NIG_TUNE_WIDE_MIN_BLOCKS) || value < -1 || value >= (int64_t)nig::TUNE_UNSET)
        return fail(NIG_ERR_INVALID, "nig_tune: unknown key or value out of range%s");
    // -1 = back to "no explicit setting": every handle uses its own device's default again
    (key == NIG_TUNE_WIDE_MIN_BLOCKS ? nig::g_wide_override : nig::g_split_override).store(value < 0 ? nig::TUNE_UNSET : (unsigned)value, std::memory_order_relaxed);
    return NIG_OK;
}
int64_t nig_tune_get(int32_t key)
{
    const unsigned cus = nig::g_last_cus.load(std::memory_order_relaxed);
    return key == NIG_TUNE_SPLIT_BLOCKS ? (int64_t)nig::split_blocks_for(cus) : key == NIG_TUNE_WIDE_MIN_BLOCKS ? (int64_t)nig::wide_min_blocks_for(cus) : -1;
}

int64_t nig_handle_tune_get(const nig_handle *h, int32_t key)
{
    if (!h) return -1;
    return key == NIG_TUNE_SPLIT_BLOCKS ? (int64_t)nig::split_blocks_for(h->cus) : key == NIG_TUNE_WIDE_MIN_BLOCKS ? (int64_t)nig::wide_min_blocks_for(h->cus) : -1;
}

int nig_clock_stamp(void *stream, uint64_t *out)
{
    if (!out) return fail(NIG_ERR_INVALID, "nig_clock_stamp: NULL buffer%s");
    hipLaunchKernelGGL(nig::clock_stamp_kernel, dim3(4096), dim3(64), 0, (hipStream_t)stream, (unsigned long long *)out);
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

int nig_env_id(const char *name)
{
    if (!name) return -1;
    for (int i = 0; i < NIG_NUM_ENVS; ++i)
        if (strcmp(name, NAMES[i]) == 0) return i;
    return -1;
}

const char *nig_env_name(int env) { return (env >= 0 && env < NIG_NUM_ENVS) ? NAMES[env] : nullptr; }

int nig_env_spec_get(int env, nig_env_spec *out)
{
    if (env < 0 || env >= NIG_NUM_ENVS || !out) return fail(NIG_ERR_INVALID, "nig_env_spec_get: bad env id%s");
    *out = SPECS[env];
    return NIG_OK;
}

static int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

int nig_layout_query(int env, int64_t batch, uint32_t flags, nig_layout *out)
{
    if (env < 0 || env >= NIG_NUM_ENVS || !out) return fail(NIG_ERR_INVALID, "nig_layout_query: bad env id%s");
    if (batch <= 0) return fail(NIG_ERR_INVALID, "nig_layout_query: batch must be positive%s");
    nig_layout L;
    L.batch = batch;
    L.ld = align_up(batch, 64);
    int64_t off = 0;
    L.off_state = off;      off = align_up(off + (int64_t)SPECS[env].state_dim * L.ld * 4, 256);
    L.off_ctr = off;        off = align_up(off + L.ld * 4, 256);
    L.off_life_viol = off;  off = align_up(off + L.ld * 8, 256);
    if (flags & NIG_F_TALLY) {
        L.off_ep_return = off;  off = align_up(off + L.ld * 8, 256);
        L.off_tally = off;      off = align_up(off + (int64_t)NIG_T_ROWS * L.ld * 8, 256);
    } else {
        L.off_ep_return = -1;
        L.off_tally = -1;
    }
    // tail: reduce scratch + the device-resident launch counter used by plans
    off = align_up(off + (int64_t)REDUCE_BLOCKS * NIG_T_ROWS * 8, 256) + 256 + POLICY_BYTES;
    L.bytes = off;
    *out = L;
    return NIG_OK;
}


int nig_create(int env, int64_t batch, int device, uint64_t seed, uint64_t env_index0, int32_t max_episode_steps,
               double dt, uint32_t flags, void *workspace, nig_handle **out)
{
    if (!out) return fail(NIG_ERR_INVALID, "nig_create: out is NULL%s");
    *out = nullptr;
    if (env < 0 || env >= NIG_NUM_ENVS) return fail(NIG_ERR_INVALID, "nig_create: unknown env id%s");
    if (batch <= 0 || batch > NIG_MAX_BATCH) return fail(NIG_ERR_INVALID, "nig_create: batch outside [1, 2^24]%s");
    if (max_episode_steps < 0 || max_episode_steps > NIG_MAX_EPISODE_STEPS)
        return fail(NIG_ERR_INVALID, "nig_create: max_episode_steps outside [1, 21845]%s");
    if (dt < 0.0 || dt != dt) return fail(NIG_ERR_INVALID, "nig_create: bad dt%s");
    if (SPECS[env].n_constraints > 3 && max_episode_steps > 16383)
        return fail(NIG_ERR_INVALID, "nig_create: max_episode_steps > 16383 for an env with 4 safety conditions%s");
    if (env == NIG_ENV_CHEMICAL_REACTOR && dt != 0.0 && dt != 0.1)
        return fail(NIG_ERR_UNSUPPORTED, "nig_create: ChemicalReactor hard-codes dt=0.1 upstream (chemical_reactor.py:68)%s");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(NIG_ERR_NODEVICE, "nig_create: no HIP device (%s); there is no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= ndev) return fail(NIG_ERR_INVALID, "nig_create: device index out of range%s");
    HIP_TRY(hipSetDevice(device));
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    nig::g_last_cus.store((unsigned)cus, std::memory_order_relaxed);

    nig_handle *h = new (std::nothrow) nig_handle();
    if (!h) return fail(NIG_ERR_INVALID, "nig_create: out of host memory%s");
    h->env = env; h->device = device; h->B = batch; h->seed = seed; h->env0 = env_index0; h->cus = (unsigned)cus;
    h->max_steps = max_episode_steps ? max_episode_steps : SPECS[env].max_episode_steps;
    h->dt = (dt != 0.0) ? dt : SPECS[env].dt;
    h->flags = flags; h->t = 0; h->cmask = 0xFu;
    nig_layout_query(env, batch, flags, &h->lay);
    if (workspace) {
        if (((uintptr_t)workspace & 255u) != 0) { delete h; return fail(NIG_ERR_INVALID, "nig_create: workspace not 256-byte aligned%s"); }
        // the step kernel updates the episode tally and the lifetime violation count with hardware atomics executed at the
        // memory side: on host-pinned / fine-grained memory they would silently miscount
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        const hipError_t pe = hipPointerGetAttributes(&at, workspace);
        // Rejected: memory POSITIVELY identified as host, managed or another device's.  A query that fails (device-local
        // memory mapped through the virtual-memory API -- hipMemCreate / hipMemMap, e.g. torch's expandable_segments allocator --
        // is not known to hipPointerGetAttributes on every ROCm release) or that reports a type this code does not know is
        // accepted: the caller has stated in the contract that it is device-local memory of `device`.
        if (pe != hipSuccess) (void)hipGetLastError();
        else if (at.type == hipMemoryTypeHost || at.type == hipMemoryTypeManaged || at.isManaged ||
                 (at.type == hipMemoryTypeDevice && at.device != device)) {
            delete h;
            return fail(NIG_ERR_INVALID, "nig_create: workspace must be ordinary device-local memory of `device` (hipMalloc / a torch CUDA tensor), "
                                         "not host-pinned, managed or another device's memory%s");
        }
        h->ws = (char *)workspace; h->owns_ws = false;
    } else {
        void *p = nullptr;
        hipError_t me = hipMalloc(&p, (size_t)h->lay.bytes);
        if (me != hipSuccess) { delete h; return fail(NIG_ERR_HIP, "hipMalloc workspace: %s", hipGetErrorString(me)); }
        h->ws = (char *)p; h->owns_ws = true;
    }
    h->pol_dev = (nig_policy *)(h->ws + h->lay.bytes - POLICY_BYTES);
    h->t_dev = (uint32_t *)(h->ws + h->lay.bytes - POLICY_BYTES - 256);
    h->scratch = (double *)(h->ws + h->lay.bytes - POLICY_BYTES - 256 - align_up((int64_t)REDUCE_BLOCKS * NIG_T_ROWS * 8, 256));
    h->has_policy = false; h->mlp_stream = nullptr; h->act32 = nullptr; h->pid_mem = nullptr; h->may_hold_done = true; h->hst_pinned = nullptr; h->hst_dev = nullptr; h->hst_bytes = 0; h->mirror = nullptr; h->ld_mirror = 0; h->act_soa = nullptr; h->act_soa_floats = 0;
    h->state = (float *)(h->ws + h->lay.off_state); h->ld_state = h->lay.ld;
    const nig_layout &L = h->lay;
    hipLaunchKernelGGL(init_ws_kernel, dim3(grid_for(L.ld)), dim3(BLOCK), 0, (hipStream_t)0,
                       (uint32_t *)(h->ws + L.off_ctr), (long long *)(h->ws + L.off_life_viol),
                       L.off_ep_return >= 0 ? (double *)(h->ws + L.off_ep_return) : nullptr,
                       L.off_tally >= 0 ? (double *)(h->ws + L.off_tally) : nullptr, L.ld, L.batch);
    hipError_t le = hipGetLastError();
    if (le == hipSuccess) le = hipMemsetAsync(h->ws + L.off_state, 0, (size_t)SPECS[env].state_dim * L.ld * 4, (hipStream_t)0);
    if (le == hipSuccess) le = hipMemsetAsync(h->t_dev, 0, 256, (hipStream_t)0);      // launch counter of plans + the ring error word of test builds
    if (le == hipSuccess) le = hipStreamSynchronize((hipStream_t)0);
    if (le != hipSuccess) {
        if (h->owns_ws) (void)hipFree(h->ws);
        delete h;
        return fail(NIG_ERR_HIP, "nig_create: workspace init failed: %s", hipGetErrorString(le));
    }
    *out = h;
    return NIG_OK;
}

int nig_destroy(nig_handle *h)
{
    if (!h) return NIG_OK;
    if (h->mlp_stream) (void)hipFree(h->mlp_stream);
    if (h->pid_mem) (void)hipFree(h->pid_mem);
    if (h->act32) (void)hipFree(h->act32);
    if (h->act_soa) (void)hipFree(h->act_soa);
    if (h->hst_dev) (void)hipFree(h->hst_dev);
    if (h->hst_pinned) (void)hipHostFree(h->hst_pinned);
    if (h->owns_ws && h->ws) (void)hipFree(h->ws);
    delete h;
    return NIG_OK;
}

int nig_get_layout(const nig_handle *h, nig_layout *out)
{
    if (!h || !out) return fail(NIG_ERR_INVALID, "nig_get_layout: NULL argument%s");
    *out = h->lay;
    return NIG_OK;
}

void *nig_workspace(const nig_handle *h) { return h ? (void *)h->ws : nullptr; }

int nig_get_counter(const nig_handle *h, uint32_t *t)
{
    if (!h || !t) return fail(NIG_ERR_INVALID, "nig_get_counter: NULL argument%s");
    *t = h->t;
    return NIG_OK;
}

int nig_bind_state(nig_handle *h, float *state, int64_t ld)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_bind_state: NULL handle%s");
    if (!state) { h->state = (float *)(h->ws + h->lay.off_state); h->ld_state = h->lay.ld; return NIG_OK; }
    if (ld < h->B || ld > NIG_MAX_PITCH) return fail(NIG_ERR_INVALID, "nig_bind_state: ld outside [batch, 2^26]%s");
    if (((uintptr_t)state & 3u) != 0) return fail(NIG_ERR_INVALID, "nig_bind_state: unaligned pointer%s");
    h->state = state; h->ld_state = ld;
    return NIG_OK;
}

int nig_set_constraint_mask(nig_handle *h, uint32_t mask)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_set_constraint_mask: NULL handle%s");
    h->cmask = mask & 0xFu;
    return NIG_OK;
}

int nig_set_counter(nig_handle *h, uint32_t t)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_set_counter: NULL handle%s");
    h->t = t;
    return NIG_OK;
}

int nig_reset(nig_handle *h, const uint8_t *mask, const double *init_noise, int64_t ld_noise, void *stream)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_reset: NULL handle%s");
    if (init_noise && ld_noise < h->B) return fail(NIG_ERR_INVALID, "nig_reset: ld_noise < batch%s");
    const nig_layout &L = h->lay;
    ResetArgs a;
    a.state = h->state; a.ld_state = h->ld_state; a.ctr = (uint32_t *)(h->ws + L.off_ctr);
    a.life_viol = (long long *)(h->ws + L.off_life_viol);
    a.ep_ret = L.off_ep_return >= 0 ? (double *)(h->ws + L.off_ep_return) : nullptr;
    a.ld = L.ld; a.B = h->B; a.mask = mask; a.noise = init_noise; a.ld_noise = ld_noise;
    a.env0 = h->env0; a.seed_lo = (uint32_t)h->seed; a.seed_hi = (uint32_t)(h->seed >> 32); a.t = h->t;
    hipStream_t st = (hipStream_t)stream;
    launch_of(h->env)->reset(a, init_noise != nullptr, grid_for(a.B), st);
    HIP_TRY(hipGetLastError());
    if (!mask) h->may_hold_done = false;          // every lane starts an episode: none is left waiting for a reset
    return NIG_OK;
}

int nig_step(nig_handle *h, const float *actions, int64_t ld_act, const double *step_noise, const double *reset_noise,
             int64_t ld_noise, float *reward_out, double *reward64_out, uint32_t *flags_out, float *final_obs,
             int64_t ld_obs, void *stream)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_step: NULL handle%s");
    if (!actions || ld_act < h->B || ld_act > NIG_MAX_PITCH) return fail(NIG_ERR_INVALID, "nig_step: actions NULL or ld_act outside [batch, 2^26]%s");
    const nig_env_spec &sp = SPECS[h->env];
    const bool autoreset = (h->flags & NIG_F_AUTORESET) != 0;
    // parity mode = the caller supplies every value the reference's RNG would have drawn
    const bool parity = (step_noise != nullptr) || (reset_noise != nullptr);
    if (parity) {
        if (sp.k_step > 0 && !step_noise) return fail(NIG_ERR_INVALID, "nig_step: parity mode needs step_noise%s");
        if (autoreset && !reset_noise) return fail(NIG_ERR_INVALID, "nig_step: parity mode with auto-reset needs reset_noise%s");
        if (ld_noise < h->B || ld_noise > NIG_MAX_PITCH) return fail(NIG_ERR_INVALID, "nig_step: ld_noise outside [batch, 2^26]%s");
    }
    if (final_obs && (ld_obs < h->B || ld_obs > NIG_MAX_PITCH)) return fail(NIG_ERR_INVALID, "nig_step: ld_obs outside [batch, 2^26]%s");
    h->t += 1;
    StepArgs a = base_step_args(h);
    a.actions = actions; a.ld_act = (uint32_t)ld_act;
    a.step_noise = step_noise; a.reset_noise = reset_noise; a.ld_noise = (uint32_t)ld_noise;
    a.reward = reward_out; a.reward64 = reward64_out; a.flags = flags_out; a.final_obs = final_obs; a.ld_obs = (uint32_t)ld_obs;
    a.t_ptr = nullptr; a.t_off = h->t;
    dispatch_step(h, a, parity, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

int nig_step64(nig_handle *h, const double *actions, int64_t ld_act, const double *step_noise, const double *reset_noise,
               int64_t ld_noise, float *reward_out, double *reward64_out, uint32_t *flags_out, float *final_obs,
               int64_t ld_obs, void *stream)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_step64: NULL handle%s");
    if (!actions || ld_act < h->B || ld_act > NIG_MAX_PITCH) return fail(NIG_ERR_INVALID, "nig_step64: actions NULL or ld_act outside [batch, 2^26]%s");
    const EnvLaunch *L = launch_of(h->env);
    if (!L->step64) {                             // this env's own arithmetic takes the action as float32
        const int A = SPECS[h->env].action_dim;
        if (!h->act32) HIP_TRY(hipMalloc((void **)&h->act32, (size_t)A * h->lay.ld * sizeof(float)));
        hipLaunchKernelGGL(narrow_rows_kernel, dim3(grid_for(h->B)), dim3(BLOCK), 0, (hipStream_t)stream, actions, ld_act,
                           h->act32, h->lay.ld, A, h->B);
        HIP_TRY(hipGetLastError());
        return nig_step(h, h->act32, h->lay.ld, step_noise, reset_noise, ld_noise, reward_out, reward64_out, flags_out,
                        final_obs, ld_obs, stream);
    }
    const nig_env_spec &sp = SPECS[h->env];
    const bool autoreset = (h->flags & NIG_F_AUTORESET) != 0;
    const bool parity = (step_noise != nullptr) || (reset_noise != nullptr);
    if (parity) {
        if (sp.k_step > 0 && !step_noise) return fail(NIG_ERR_INVALID, "nig_step64: parity mode needs step_noise%s");
        if (autoreset && !reset_noise) return fail(NIG_ERR_INVALID, "nig_step64: parity mode with auto-reset needs reset_noise%s");
        if (ld_noise < h->B || ld_noise > NIG_MAX_PITCH) return fail(NIG_ERR_INVALID, "nig_step64: ld_noise outside [batch, 2^26]%s");
    }
    if (final_obs && (ld_obs < h->B || ld_obs > NIG_MAX_PITCH)) return fail(NIG_ERR_INVALID, "nig_step64: ld_obs outside [batch, 2^26]%s");
    h->t += 1;
    StepArgs a = base_step_args(h);
    a.actions64 = actions; a.ld_act = (uint32_t)ld_act;
    a.step_noise = step_noise; a.reset_noise = reset_noise; a.ld_noise = (uint32_t)ld_noise;
    a.reward = reward_out; a.reward64 = reward64_out; a.flags = flags_out; a.final_obs = final_obs; a.ld_obs = (uint32_t)ld_obs;
    a.t_ptr = nullptr; a.t_off = h->t;
    L->step64(a, parity, grid_for(h->B), (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

static int rollout_impl(nig_handle *h, int32_t n_steps, const float *action_ring, int64_t ld_act, int64_t slot_stride,
                        int32_t ring_len, const double *step_noise, int64_t step_noise_stride, const double *reset_noise,
                        int64_t reset_noise_stride, int64_t ld_noise, float *reward_out, uint32_t *flags_out, int64_t out_stride,
                        float *obs_out, int64_t ld_obs, int64_t obs_step_stride, void *stream)
{
    if (!h || !action_ring || n_steps <= 0 || ring_len <= 0) return fail(NIG_ERR_INVALID, "nig_rollout: bad argument%s");
    // ld_act == 0: the ring is ROW-MAJOR, slot s = [B][A] at action_ring + s * slot_stride (what an agent's batched output looks like)
    const bool act_aos = ld_act == 0;
    const int A_ = SPECS[h->env].action_dim;
    if (act_aos) {
        if (step_noise || reset_noise) return fail(NIG_ERR_UNSUPPORTED, "nig_rollout_noise: the recorded-draw launches take [A][ld_act] action rows%s");
        if (slot_stride < (int64_t)A_ * h->B || slot_stride > 0xffffffffLL)
            return fail(NIG_ERR_INVALID, "nig_rollout: row-major action ring: slot_stride smaller than one [batch][A] slot (or >= 2^32)%s");
    } else {
    if (ld_act < h->B || ld_act > NIG_MAX_PITCH) return fail(NIG_ERR_INVALID, "nig_rollout: ld_act outside {0} U [batch, 2^26]%s");
    if (slot_stride < (int64_t)A_ * ld_act || slot_stride > 0xffffffffLL)
        return fail(NIG_ERR_INVALID, "nig_rollout: slot_stride smaller than one [A][ld_act] slot (or >= 2^32)%s");
    }
    if (out_stride != 0 && (out_stride < h->B || out_stride > NIG_MAX_PITCH))
        return fail(NIG_ERR_INVALID, "nig_rollout: out_stride outside {0} U [batch, 2^26]%s");
    if ((int64_t)n_steps * out_stride > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout: n_steps*out_stride >= 2^32%s");
    const bool obs_aos = obs_out && ld_obs == 0;
    if (obs_out && !obs_aos && (ld_obs < h->B || ld_obs > NIG_MAX_PITCH ||
                                (obs_step_stride != 0 && obs_step_stride < (int64_t)SPECS[h->env].state_dim * ld_obs)))
        return fail(NIG_ERR_INVALID, "nig_rollout: bad observation trajectory pitch%s");
    if (obs_aos && ((obs_step_stride != 0 && obs_step_stride < (int64_t)SPECS[h->env].state_dim * h->B) || (obs_step_stride & 3) || ((uintptr_t)obs_out & 15)))
        return fail(NIG_ERR_INVALID, "nig_rollout: row-major trajectory needs 16-byte alignment and obs_step_stride 0 or >= S*batch (multiple of 4)%s");
    if ((int64_t)h->t + n_steps > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout: launch counter would wrap%s");
    RolloutArgs q;
    memset(&q, 0, sizeof q);
    q.s = base_step_args(h);
    q.s.actions = action_ring; q.s.ld_act = (uint32_t)ld_act;
    q.s.reward = reward_out; q.s.flags = flags_out;
    q.s.t_ptr = nullptr; q.s.t_off = h->t;
    q.n_steps = n_steps; q.it0 = 0; q.ring_len = ring_len; q.slot_stride = (uint32_t)slot_stride; q.out_stride = (uint32_t)out_stride;
    q.obs_out = obs_out; q.ld_obs_out = (uint32_t)ld_obs; q.obs_step_stride = (uint64_t)obs_step_stride; q.obs_aos = obs_aos ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    if ((reward_out == nullptr) != (flags_out == nullptr))
        return fail(NIG_ERR_INVALID, "nig_rollout: reward_out and flags_out go together (both or neither)%s");
    if (obs_out && !reward_out)
        return fail(NIG_ERR_INVALID, "nig_rollout: an observation trajectory needs reward_out and flags_out too%s");
    const int out_mode = !reward_out ? 0 : (!obs_out ? 1 : (obs_aos ? 3 : 2));
    if (act_aos) {
        // NATIVE where every kernel of the launch reads a lane's actions as contiguous bytes: PowerGrid's LDS-resident body
        // (csrc/nig_pg_lds.hpp: two 16-byte loads per lane, 2 KiB contiguous per wave instead of eight 256-byte row segments) in its
        // wide 512 / 256 and paired forms -- the launcher's own predicate (rollout_rows_native, next to launch_rollout_form) -- and
        // 16-byte aligned slots.  Every other launch reads [A][ld] rows: the ring is transposed into a buffer the handle owns, on
        // the caller's stream, per call.
        const bool native = launch_of(h->env)->rows_native(out_mode, q) && ((uintptr_t)action_ring & 15) == 0 && (slot_stride & 3) == 0;
        if (!native) {
            if ((int64_t)A_ * h->lay.ld > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout: row-major action ring: batch too large for the row copy%s");
            const int used = ring_len < n_steps ? ring_len : n_steps;     // the slots this call reads: step k takes slot k % ring_len, from 0
            const size_t need = (size_t)used * A_ * (size_t)h->lay.ld;
            if (h->act_soa_floats < need) {
                if (h->act_soa) { HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(h->act_soa); h->act_soa = nullptr; h->act_soa_floats = 0; }
                HIP_TRY(hipMalloc((void **)&h->act_soa, need * sizeof(float)));
                h->act_soa_floats = need;
            }
            hipLaunchKernelGGL(nig::action_rows_kernel, dim3(grid_for(h->B), (unsigned)(used < 65535 ? used : 65535)), dim3(BLOCK), 0, st,
                               action_ring, slot_stride, h->act_soa, h->lay.ld, h->B, A_, used);
            HIP_TRY(hipGetLastError());
            q.s.actions = h->act_soa; q.s.ld_act = (uint32_t)h->lay.ld; q.slot_stride = (uint32_t)((int64_t)A_ * h->lay.ld);
        }
    }
    if (step_noise || reset_noise) {               // nig_rollout_noise: the reference's recorded draws
        const nig_env_spec &sp = SPECS[h->env];
        if (h->env > NIG_ENV_ROBOT_ASSEMBLY)
            return fail(NIG_ERR_UNSUPPORTED, "nig_rollout_noise: only the envs the reference can record draws for (ChemicalReactor, PowerGrid, RobotAssembly)%s");
        if (out_mode != 3)
            return fail(NIG_ERR_UNSUPPORTED, "nig_rollout_noise: the injected-draw kernels exist for the row-major trajectory (obs_out with ld_obs == 0) only%s");
        if (ring_len < n_steps) return fail(NIG_ERR_INVALID, "nig_rollout_noise: recorded draws belong to recorded actions: ring_len >= n_steps%s");
        if (ld_noise < h->B || ld_noise > NIG_MAX_PITCH) return fail(NIG_ERR_INVALID, "nig_rollout_noise: ld_noise outside [batch, 2^26]%s");
        if (sp.k_step > 0 && (!step_noise || step_noise_stride < (int64_t)sp.k_step * ld_noise))
            return fail(NIG_ERR_INVALID, "nig_rollout_noise: step_noise NULL or its step stride smaller than one [k_step][ld_noise] row set%s");
        if ((h->flags & NIG_F_AUTORESET) != 0 && sp.k_reset > 0 && (!reset_noise || reset_noise_stride < (int64_t)sp.k_reset * ld_noise))
            return fail(NIG_ERR_INVALID, "nig_rollout_noise: an auto-reset handle needs reset_noise, step stride >= one [k_reset][ld_noise] row set%s");
        q.s.step_noise = sp.k_step > 0 ? step_noise : nullptr;
        q.s.reset_noise = reset_noise;
        if (!q.s.step_noise && !q.s.reset_noise) q.s.reset_noise = reset_noise ? reset_noise : step_noise;   // (keeps the launcher on the injected-draw path)
        q.s.ld_noise = (uint32_t)ld_noise;
        q.nz_step_stride = (uint64_t)step_noise_stride; q.nz_reset_stride = (uint64_t)reset_noise_stride;
    }
    launch_of(h->env)->rollout(out_mode, q, h->t + 1u, grid_for(h->B), st);
    HIP_TRY(hipGetLastError());
    h->t += (uint32_t)n_steps;
    return ring_check(h, st, "nig_rollout");
}

int nig_rollout(nig_handle *h, int32_t n_steps, const float *action_ring, int64_t ld_act, int64_t slot_stride,
                int32_t ring_len, float *reward_out, uint32_t *flags_out, int64_t out_stride,
                float *obs_out, int64_t ld_obs, int64_t obs_step_stride, void *stream)
{
    return rollout_impl(h, n_steps, action_ring, ld_act, slot_stride, ring_len, nullptr, 0, nullptr, 0, 0, reward_out, flags_out,
                        out_stride, obs_out, ld_obs, obs_step_stride, stream);
}

int nig_rollout_noise(nig_handle *h, int32_t n_steps, const float *action_ring, int64_t ld_act, int64_t slot_stride,
                      int32_t ring_len, const double *step_noise, int64_t step_noise_stride, const double *reset_noise,
                      int64_t reset_noise_stride, int64_t ld_noise, float *reward_out, uint32_t *flags_out, int64_t out_stride,
                      float *obs_out, int64_t obs_step_stride, void *stream)
{
    if (!step_noise && !reset_noise) return fail(NIG_ERR_INVALID, "nig_rollout_noise: no recorded draws given (use nig_rollout)%s");
    return rollout_impl(h, n_steps, action_ring, ld_act, slot_stride, ring_len, step_noise, step_noise_stride, reset_noise,
                        reset_noise_stride, ld_noise, reward_out, flags_out, out_stride, obs_out, 0, obs_step_stride, stream);
}

int nig_set_policy(nig_handle *h, const nig_policy *policy, void *stream)
{
    if (!h || !policy) return fail(NIG_ERR_INVALID, "nig_set_policy: NULL argument%s");
    static_assert(sizeof(nig_policy) <= POLICY_BYTES, "policy struct outgrew its device slot");
    if (policy->kind != NIG_POLICY_AFFINE && policy->kind != NIG_POLICY_PID)
        return fail(NIG_ERR_INVALID, "nig_set_policy: unknown policy kind%s");
    if (!(policy->clip_lo <= policy->clip_hi)) return fail(NIG_ERR_INVALID, "nig_set_policy: clip_lo > clip_hi%s");
    h->pol_host = *policy;
    // recompute the non-zero column mask here so a caller cannot get it wrong
    uint32_t cm = 0;
    for (int k = 0; k < SPECS[h->env].state_dim; ++k)
        for (int j = 0; j < SPECS[h->env].action_dim; ++j)
            if (policy->Wt[k][j] != 0.0f) cm |= (1u << k);
    h->pol_host.colmask = cm;
    if (policy->kind == NIG_POLICY_PID) {         // the agent's constructor: integral = previous_error = 0 (baseline_agents.py:55-57)
        const size_t bytes = (size_t)2 * SPECS[h->env].action_dim * h->lay.ld * sizeof(float);
        if (!h->pid_mem) HIP_TRY(hipMalloc((void **)&h->pid_mem, bytes));
        HIP_TRY(hipMemsetAsync(h->pid_mem, 0, bytes, (hipStream_t)stream));
    }
    HIP_TRY(hipMemcpyAsync(h->pol_dev, &h->pol_host, sizeof(nig_policy), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));   // pageable source: make the staging copy reusable
    h->has_policy = true;
    return NIG_OK;
}

int nig_rollout_policy(nig_handle *h, int32_t n_steps, float *reward_out, uint32_t *flags_out, int64_t out_stride,
                       float *obs_out, int64_t obs_step_stride, float *act_out, int64_t ld_act,
                       int64_t act_step_stride, void *stream)
{
    if (!h || n_steps <= 0) return fail(NIG_ERR_INVALID, "nig_rollout_policy: bad argument%s");
    if (!h->has_policy) return fail(NIG_ERR_INVALID, "nig_rollout_policy: no policy installed (nig_set_policy)%s");
    if (out_stride != 0 && (out_stride < h->B || out_stride > NIG_MAX_PITCH))
        return fail(NIG_ERR_INVALID, "nig_rollout_policy: out_stride outside {0} U [batch, 2^26]%s");
    if ((int64_t)n_steps * out_stride > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout_policy: n_steps*out_stride >= 2^32%s");
    if (obs_out && (obs_step_stride < (int64_t)SPECS[h->env].state_dim * h->B || (obs_step_stride & 3) || ((uintptr_t)obs_out & 15)))
        return fail(NIG_ERR_INVALID, "nig_rollout_policy: obs_out needs 16-byte alignment and obs_step_stride >= S*batch (multiple of 4)%s");
    if (act_out && (ld_act < h->B || ld_act > NIG_MAX_PITCH || act_step_stride < (int64_t)SPECS[h->env].action_dim * ld_act))
        return fail(NIG_ERR_INVALID, "nig_rollout_policy: bad action trajectory pitch%s");
    if ((int64_t)h->t + n_steps > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout_policy: launch counter would wrap%s");
    PolicyArgs q;
    memset(&q, 0, sizeof q);
    q.s = base_step_args(h);
    q.s.reward = reward_out; q.s.flags = flags_out;
    q.s.t_ptr = nullptr; q.s.t_off = h->t;
    q.pol = h->pol_dev; q.n_steps = n_steps; q.out_stride = (uint32_t)out_stride;
    q.pid = (h->pol_host.kind == NIG_POLICY_PID) ? h->pid_mem : nullptr;
    q.obs_out = obs_out; q.obs_step_stride = (uint64_t)obs_step_stride;
    q.act_out = act_out; q.ld_act_out = (uint32_t)ld_act; q.act_step_stride = (uint64_t)act_step_stride;
    q.pol_kind = h->pol_host.kind;
    hipStream_t st = (hipStream_t)stream;
    launch_of(h->env)->policy(q, grid_for(h->B), st);
    HIP_TRY(hipGetLastError());
    h->t += (uint32_t)n_steps;
    return ring_check(h, st, "nig_rollout_policy");
}

// Row of a 32x32 MFMA result tile held in register t by lane half hf (MI355X_MICROARCH / guide section 3).
static inline int mfma_row(int t, int hf) { return (t & 3) + 8 * (t >> 2) + 4 * hf; }

int nig_set_mlp_policy(nig_handle *h, int32_t hidden, const float *W1, const float *b1, const float *W2, const float *b2,
                       const float *W3, const float *b3, void *stream)
{
    if (!h || !W1 || !b1 || !W2 || !b2 || !W3 || !b3) return fail(NIG_ERR_INVALID, "nig_set_mlp_policy: NULL argument%s");
    if (hidden != MLP_H) return fail(NIG_ERR_UNSUPPORTED, "nig_set_mlp_policy: hidden must be 256 (agents/networks.py default)%s");
    const int S = SPECS[h->env].state_dim, A = SPECS[h->env].action_dim, H = MLP_H;
    if (S % 2 != 0 || A > 16) return fail(NIG_ERR_UNSUPPORTED, "nig_set_mlp_policy: env shape not supported (even state dim, at most 16 actions)%s");
    float *host = (float *)calloc((size_t)MLP_STREAM_FLOATS, sizeof(float));
    if (!host) return fail(NIG_ERR_INVALID, "nig_set_mlp_policy: out of host memory%s");
    // Build the operand stream in exactly the order rollout_mlp_kernel consumes it, chunk by chunk (a chunk = one
    // fill of an LDS buffer, MLP_CHREC record slots, zero padded).  Record = 64 floats; lane l = (i = l & 31,
    // hf = l >> 5) holds W[k(hf)][32*tile + i].
    auto rec = [&](int chunk, int r) { return host + ((size_t)chunk * MLP_CHREC + r) * 64; };
    for (int m = 0; m < MLP_MT; ++m) {                      // chunk 0: layer 1, natural k order: k = 2*ks + hf
        const int R1 = S / 2 + 1;
        for (int ks = 0; ks < S / 2; ++ks)
            for (int l = 0; l < 64; ++l) rec(0, m * R1 + ks)[l] = W1[(size_t)(2 * ks + (l >> 5)) * H + 32 * m + (l & 31)];
        for (int l = 0; l < 32; ++l) rec(0, m * R1 + S / 2)[l] = b1[32 * m + l];
    }
    for (int m2 = 0; m2 < MLP_MT; ++m2) {                   // chunk 1 + m2
        int r = 0;
        for (int kt = 0; kt < MLP_MT; ++kt)                   // layer 2: k follows the accumulator register order of h1
            for (int t = 0; t < 16; ++t, ++r)
                for (int l = 0; l < 64; ++l)
                    rec(1 + m2, r)[l] = W2[(size_t)(32 * kt + mfma_row(t, l >> 5)) * H + 32 * m2 + (l & 31)];
        for (int l = 0; l < 32; ++l) rec(1 + m2, r)[l] = b2[32 * m2 + l];
        ++r;
        for (int t = 0; t < 16; ++t, ++r)                     // head: rows i >= A are zero
            for (int l = 0; l < 64; ++l) {
                if (A <= 4) {                                 // v_mfma_f32_4x4x1: lane 4 b + i of every 4-lane block holds head row i of ITS half's hidden row
                    if ((l & 3) < A) rec(1 + m2, r)[l] = W3[(size_t)(32 * m2 + mfma_row(t, l >> 5)) * A + (l & 3)];
                } else if ((l & 15) < A)                      // v_mfma_f32_16x16x1 (four blocks): lane 16 b + i holds head row i of block b's hidden row
                    rec(1 + m2, r)[l] = W3[(size_t)(32 * m2 + mfma_row(t, l >> 5)) * A + (l & 15)];
            }
        if (r != MLP_PER) { free(host); return fail(NIG_ERR_INVALID, "nig_set_mlp_policy: internal record count mismatch%s"); }
    }
    if (A <= 4) {                                                    // the head's bias rides at the end of the last chunk
        for (int l = 0; l < 64; ++l)
            if ((l & 3) < A) rec(MLP_MT, MLP_PER)[l] = b3[l & 3];    // (4 x 4 x 1: every block's row lanes; B = 1 on lane half 0, 0 on half 1)
    } else
        for (int l = 0; l < 64; ++l)
            if ((l & 15) < A) rec(MLP_MT, MLP_PER)[l] = b3[l & 15];
    hipError_t e = hipSuccess;
    if (!h->mlp_stream) e = hipMalloc((void **)&h->mlp_stream, (size_t)MLP_STREAM_FLOATS * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(h->mlp_stream, host, (size_t)MLP_STREAM_FLOATS * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    if (e != hipSuccess) return fail(NIG_ERR_HIP, "nig_set_mlp_policy: %s", hipGetErrorString(e));
    return NIG_OK;
}

int nig_rollout_mlp(nig_handle *h, int32_t n_steps, float *reward_out, uint32_t *flags_out, int64_t out_stride,
                    float *obs_out, int64_t obs_step_stride, float *act_out, int64_t ld_act, int64_t act_step_stride,
                    void *stream)
{
    if (!h || n_steps <= 0) return fail(NIG_ERR_INVALID, "nig_rollout_mlp: bad argument%s");
    if (!h->mlp_stream || !launch_of(h->env)->mlp) return fail(NIG_ERR_INVALID, "nig_rollout_mlp: no actor installed (nig_set_mlp_policy)%s");
    if (out_stride != 0 && (out_stride < h->B || out_stride > NIG_MAX_PITCH))
        return fail(NIG_ERR_INVALID, "nig_rollout_mlp: out_stride outside {0} U [batch, 2^26]%s");
    if ((int64_t)n_steps * out_stride > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout_mlp: n_steps*out_stride >= 2^32%s");
    if (obs_out && (obs_step_stride < (int64_t)SPECS[h->env].state_dim * h->B || (obs_step_stride & 3) || ((uintptr_t)obs_out & 15)))
        return fail(NIG_ERR_INVALID, "nig_rollout_mlp: obs_out needs 16-byte alignment and obs_step_stride >= S*batch (multiple of 4)%s");
    if (act_out && (ld_act < h->B || ld_act > NIG_MAX_PITCH || act_step_stride < (int64_t)SPECS[h->env].action_dim * ld_act))
        return fail(NIG_ERR_INVALID, "nig_rollout_mlp: bad action trajectory pitch%s");
    if ((int64_t)h->t + n_steps > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout_mlp: launch counter would wrap%s");
    MlpArgs q;
    memset(&q, 0, sizeof q);
    q.s = base_step_args(h);
    q.s.reward = reward_out; q.s.flags = flags_out;
    q.s.t_ptr = nullptr; q.s.t_off = h->t;
    q.wstream = h->mlp_stream; q.n_steps = n_steps; q.out_stride = (uint32_t)out_stride;
    q.obs_out = obs_out; q.obs_step_stride = (uint64_t)obs_step_stride;
    q.act_out = act_out; q.ld_act_out = (uint32_t)ld_act; q.act_step_stride = (uint64_t)act_step_stride;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((h->B + BLOCK / 2 - 1) / (BLOCK / 2));     // 32 envs per wave, 128 per block
    launch_of(h->env)->mlp(q, grid, st);
    HIP_TRY(hipGetLastError());
    h->t += (uint32_t)n_steps;
    return NIG_OK;
}

// ---- host-buffer entry points (small batches) -------------------------------------------------
// staging layout (same on host and device): [actions f32 A*B][noise f64 K*B] | [state f32 S*B][reward64 f64 B][flags u32 B]
// (uploads = the first part, downloads = the second part: one memcpy each way)
struct HostStage { size_t off_act, off_noise, off_state, off_rew, off_flags, bytes; };

static HostStage host_stage_layout(const nig_handle *h)
{
    const nig_env_spec &sp = SPECS[h->env];
    const size_t B = (size_t)h->B;
    const size_t K = (size_t)(sp.k_step > sp.k_reset ? sp.k_step : sp.k_reset);
    HostStage L;
    L.off_act = 0;
    L.off_noise = (size_t)align_up((int64_t)(sp.action_dim * B * 8), 256);     // float32 or float64 action rows
    L.off_state = L.off_noise + (size_t)align_up((int64_t)(K * B * 8), 256);
    L.off_rew = L.off_state + (size_t)align_up((int64_t)(sp.state_dim * B * 4), 256);
    L.off_flags = L.off_rew + (size_t)align_up((int64_t)(B * 8), 256);
    L.bytes = L.off_flags + (size_t)align_up((int64_t)(B * 4), 256);
    return L;
}

constexpr int64_t HOST_ZERO_COPY_MAX = 1024;   // lanes up to which the host-buffer entry points skip the staging copies

static int host_stage_ensure(nig_handle *h, const HostStage &L)
{
    if (h->hst_pinned && h->hst_bytes >= L.bytes) return NIG_OK;
    if (h->B > 65536) return fail(NIG_ERR_UNSUPPORTED, "host-buffer entry points are for small batches (<= 65536 lanes)%s");
    if (h->hst_dev) { (void)hipFree(h->hst_dev); h->hst_dev = nullptr; }
    if (h->hst_pinned) { (void)hipHostFree(h->hst_pinned); h->hst_pinned = nullptr; }
    HIP_TRY(hipHostMalloc((void **)&h->hst_pinned, L.bytes, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **)&h->hst_dev, L.bytes));
    h->hst_bytes = L.bytes;
    return NIG_OK;
}

int nig_reset_host(nig_handle *h, const double *init_noise, float *state_out, void *stream)
{
    if (!h || !state_out) return fail(NIG_ERR_INVALID, "nig_reset_host: NULL argument%s");
    const nig_env_spec &sp = SPECS[h->env];
    const HostStage L = host_stage_layout(h);
    int rc = host_stage_ensure(h, L);
    if (rc != NIG_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t B = (size_t)h->B;
    const double *dn = nullptr;
    if (init_noise && sp.k_reset > 0) {
        memcpy(h->hst_pinned + L.off_noise, init_noise, (size_t)sp.k_reset * B * 8);
        HIP_TRY(hipMemcpyAsync(h->hst_dev + L.off_noise, h->hst_pinned + L.off_noise, (size_t)sp.k_reset * B * 8, hipMemcpyHostToDevice, st));
        dn = (const double *)(h->hst_dev + L.off_noise);
    }
    rc = nig_reset(h, nullptr, dn, (int64_t)B, stream);
    if (rc != NIG_OK) return rc;
    // gather the rows (ld apart on the device) into the contiguous staging image, then ONE download
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(h->B)), dim3(BLOCK), 0, st, (const float *)h->state, h->ld_state,
                       (float *)(h->hst_dev + L.off_state), (int64_t)B, sp.state_dim, h->B);
    HIP_TRY(hipMemcpyAsync(h->hst_pinned + L.off_state, h->hst_dev + L.off_state, (size_t)sp.state_dim * B * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(state_out, h->hst_pinned + L.off_state, (size_t)sp.state_dim * B * 4);
    return NIG_OK;
}

static int step_host_impl(nig_handle *h, const void *actions, bool act64, const double *step_noise, float *state_out,
                          double *reward64_out, uint32_t *flags_out, void *stream);

int nig_step_host(nig_handle *h, const float *actions, const double *step_noise, float *state_out, double *reward64_out,
                  uint32_t *flags_out, void *stream)
{
    return step_host_impl(h, actions, false, step_noise, state_out, reward64_out, flags_out, stream);
}

int nig_step_host64(nig_handle *h, const double *actions, const double *step_noise, float *state_out, double *reward64_out,
                    uint32_t *flags_out, void *stream)
{
    return step_host_impl(h, actions, true, step_noise, state_out, reward64_out, flags_out, stream);
}

static int step_host_impl(nig_handle *h, const void *actions, bool act64, const double *step_noise, float *state_out,
                          double *reward64_out, uint32_t *flags_out, void *stream)
{
    if (!h || !actions || !state_out || !reward64_out || !flags_out) return fail(NIG_ERR_INVALID, "nig_step_host: NULL argument%s");
    const nig_env_spec &sp = SPECS[h->env];
    const HostStage L = host_stage_layout(h);
    int rc = host_stage_ensure(h, L);
    if (rc != NIG_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t B = (size_t)h->B;
    const size_t act_bytes = (size_t)sp.action_dim * B * (act64 ? 8 : 4);
    memcpy(h->hst_pinned + L.off_act, actions, act_bytes);
    size_t up = act_bytes;
    const double *dn = nullptr;
    if (step_noise && sp.k_step > 0) {
        memcpy(h->hst_pinned + L.off_noise, step_noise, (size_t)sp.k_step * B * 8);
        up = L.off_noise + (size_t)sp.k_step * B * 8;          // one contiguous upload covers both
        dn = (const double *)(h->hst_dev + L.off_noise);
    }
    // Small batches (the single-env drop-in class is B = 1): the kernels read the actions / noise from, and
    // write state / reward / flags to, the pinned staging buffer itself -- hipHostMalloc memory is mapped
    // and coherent, a few hundred bytes over PCIe cost less than two copy commands -- so a step is two
    // kernel launches and one stream sync.  Larger batches keep one upload + one download.
    const bool zero_copy = h->B <= HOST_ZERO_COPY_MAX;
    char *io = zero_copy ? h->hst_pinned : h->hst_dev;
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(h->hst_dev, h->hst_pinned, up, hipMemcpyHostToDevice, st));
    if (dn) dn = (const double *)(io + L.off_noise);
    h->mirror = (float *)(io + L.off_state); h->ld_mirror = (uint32_t)B;
    rc = act64 ? nig_step64(h, (const double *)(io + L.off_act), (int64_t)B, dn, nullptr, (int64_t)B, nullptr,
                            (double *)(io + L.off_rew), (uint32_t *)(io + L.off_flags), nullptr, 0, stream)
               : nig_step(h, (const float *)(io + L.off_act), (int64_t)B, dn, nullptr, (int64_t)B, nullptr,
                          (double *)(io + L.off_rew), (uint32_t *)(io + L.off_flags), nullptr, 0, stream);
    h->mirror = nullptr;
    if (rc != NIG_OK) return rc;
    // (the state rows land next to reward64 and flags -- one download for everything the call returns -- written by the step
    // kernel itself: StepArgs::mirror.  Until round 4 a row-gather kernel was launched behind every step for them.)
    if (!zero_copy) HIP_TRY(hipMemcpyAsync(h->hst_pinned + L.off_state, h->hst_dev + L.off_state, L.bytes - L.off_state, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(state_out, h->hst_pinned + L.off_state, (size_t)sp.state_dim * B * 4);
    memcpy(reward64_out, h->hst_pinned + L.off_rew, B * 8);
    memcpy(flags_out, h->hst_pinned + L.off_flags, B * 4);
    return NIG_OK;
}

int nig_plan_create(nig_handle *h, int32_t n_steps, const float *action_ring, int64_t ld_act, int64_t slot_stride,
                    int32_t ring_len, float *reward_out, uint32_t *flags_out, int64_t out_stride, nig_plan **out)
{
    if (!out) return fail(NIG_ERR_INVALID, "nig_plan_create: out is NULL%s");
    *out = nullptr;
    if (!h || !action_ring || n_steps <= 0 || ring_len <= 0 || ld_act < h->B || ld_act > NIG_MAX_PITCH)
        return fail(NIG_ERR_INVALID, "nig_plan_create: bad argument%s");
    if (slot_stride < (int64_t)SPECS[h->env].action_dim * ld_act)
        return fail(NIG_ERR_INVALID, "nig_plan_create: slot_stride smaller than one [A][ld_act] slot%s");
    if (out_stride != 0 && out_stride < h->B) return fail(NIG_ERR_INVALID, "nig_plan_create: out_stride < batch%s");
    nig_plan *p = new (std::nothrow) nig_plan();
    if (!p) return fail(NIG_ERR_INVALID, "nig_plan_create: out of host memory%s");
    p->h = h; p->n_steps = n_steps; p->graph = nullptr; p->exec = nullptr;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t cs = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    if (e != hipSuccess) { delete p; return fail(NIG_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    e = hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed);
    if (e == hipSuccess) {
        for (int k = 0; k < n_steps; ++k) {
            StepArgs a = base_step_args(h);
            const int slot = k % ring_len;
            a.actions = action_ring + (int64_t)slot * slot_stride; a.ld_act = (uint32_t)ld_act;
            a.reward = reward_out ? reward_out + (int64_t)slot * out_stride : nullptr;
            a.flags = flags_out ? flags_out + (int64_t)slot * out_stride : nullptr;
            a.t_ptr = h->t_dev; a.t_off = (uint32_t)(k + 1);
            dispatch_step(h, a, false, cs);
        }
        e = hipStreamEndCapture(cs, &p->graph);
    }
    if (e == hipSuccess) e = hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0);
    (void)hipStreamDestroy(cs);
    if (e != hipSuccess) {
        if (p->graph) (void)hipGraphDestroy(p->graph);
        delete p;
        return fail(NIG_ERR_HIP, "nig_plan_create: graph capture failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return NIG_OK;
}

int nig_plan_launch(nig_plan *p, void *stream)
{
    if (!p) return fail(NIG_ERR_INVALID, "nig_plan_launch: NULL plan%s");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(set_u32_kernel, dim3(1), dim3(1), 0, st, p->h->t_dev, p->h->t);
    HIP_TRY(hipGraphLaunch(p->exec, st));
    p->h->t += (uint32_t)p->n_steps;
    return NIG_OK;
}

int nig_plan_destroy(nig_plan *p)
{
    if (!p) return NIG_OK;
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    delete p;
    return NIG_OK;
}

// ---- mixed batch -------------------------------------------------------------------------------
struct nig_mixed {
    int n;
    int device;
    nig_handle *seg[NIG_MIXED_MAX_SEGMENTS];
    int64_t off[NIG_MIXED_MAX_SEGMENTS];
    int64_t ld, lanes;
    int s_max, a_max;
    float *state;              // [s_max][ld], owned
};

// relative cost of one env-step (fused rollout, measured per-env rates): launch order = most expensive first.  (Round 4,
// profiles/r04/mixed_order.txt: PowerGrid before RobotAssembly or after makes no difference, 2.69 vs 2.70 ms; the cheap
// envs first costs +10 %, RobotAssembly last +13 %.)
static int env_cost(int env)
{
    static const int C[NIG_NUM_ENVS] = {10, 44, 50, 15, 20, 14, 9, 17, 29};
    return C[env];
}

int nig_rollout_mixed(nig_handle *const *handles, const int64_t *lane_offsets, int32_t n_handles, int32_t n_steps,
                      const float *action_ring, int64_t ld_act, int64_t slot_stride, int32_t ring_len,
                      float *reward_out, uint32_t *flags_out, int64_t out_stride, void *stream)
{
    return nig_rollout_mixed_obs(handles, lane_offsets, n_handles, n_steps, action_ring, ld_act, slot_stride, ring_len, reward_out,
                                 flags_out, out_stride, nullptr, 0, 0, stream);
}

int nig_rollout_mixed_obs(nig_handle *const *handles, const int64_t *lane_offsets, int32_t n_handles, int32_t n_steps,
                          const float *action_ring, int64_t ld_act, int64_t slot_stride, int32_t ring_len,
                          float *reward_out, uint32_t *flags_out, int64_t out_stride,
                          float *obs_out, int64_t ld_obs, int64_t obs_step_stride, void *stream)
{
    if (!handles || !lane_offsets || n_handles <= 0 || n_handles > NIG_MIXED_MAX_SEGMENTS)
        return fail(NIG_ERR_INVALID, "nig_rollout_mixed: 1..12 handles%s");
    if (!action_ring || n_steps <= 0 || ring_len <= 0) return fail(NIG_ERR_INVALID, "nig_rollout_mixed: bad argument%s");
    if ((reward_out == nullptr) != (flags_out == nullptr))
        return fail(NIG_ERR_INVALID, "nig_rollout_mixed: reward_out and flags_out go together (both or neither)%s");
    if (ld_act <= 0 || ld_act > NIG_MAX_PITCH || slot_stride > 0xffffffffLL)
        return fail(NIG_ERR_INVALID, "nig_rollout_mixed: ld_act outside [1, 2^26] or slot_stride >= 2^32%s");
    if (out_stride != 0 && out_stride > NIG_MAX_PITCH) return fail(NIG_ERR_INVALID, "nig_rollout_mixed: out_stride > 2^26%s");
    if ((int64_t)n_steps * out_stride > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout_mixed: n_steps*out_stride >= 2^32%s");
    if (obs_out && !reward_out) return fail(NIG_ERR_INVALID, "nig_rollout_mixed: an observation trajectory needs reward_out and flags_out too%s");
    if (obs_out && (ld_obs <= 0 || ld_obs > NIG_MAX_PITCH)) return fail(NIG_ERR_INVALID, "nig_rollout_mixed: ld_obs outside [1, 2^26]%s");
    int order[NIG_MIXED_MAX_SEGMENTS];
    int a_max = 0, s_max = 0;
    for (int k = 0; k < n_handles; ++k) {
        const nig_handle *h = handles[k];
        if (!h) return fail(NIG_ERR_INVALID, "nig_rollout_mixed: NULL handle%s");
        if (h->device != handles[0]->device) return fail(NIG_ERR_INVALID, "nig_rollout_mixed: handles on different devices%s");
        if (lane_offsets[k] < 0 || lane_offsets[k] + h->B > ld_act || (out_stride != 0 && lane_offsets[k] + h->B > out_stride))
            return fail(NIG_ERR_INVALID, "nig_rollout_mixed: a segment does not fit the row pitch%s");
        if ((int64_t)h->t + n_steps > 0xffffffffLL) return fail(NIG_ERR_INVALID, "nig_rollout_mixed: launch counter would wrap%s");
        if (SPECS[h->env].action_dim > a_max) a_max = SPECS[h->env].action_dim;
        if (SPECS[h->env].state_dim > s_max) s_max = SPECS[h->env].state_dim;
        if (obs_out && lane_offsets[k] + h->B > ld_obs) return fail(NIG_ERR_INVALID, "nig_rollout_mixed: a segment does not fit ld_obs%s");
        order[k] = k;
    }
    if (obs_out && obs_step_stride != 0 && obs_step_stride < (int64_t)s_max * ld_obs)
        return fail(NIG_ERR_INVALID, "nig_rollout_mixed: obs_step_stride smaller than one [S_max][ld_obs] step%s");
    if (slot_stride < (int64_t)a_max * ld_act) return fail(NIG_ERR_INVALID, "nig_rollout_mixed: slot_stride smaller than one [A_max][ld_act] slot%s");
    for (int i = 1; i < n_handles; ++i)           // insertion sort: most expensive env first, ties in segment order
        for (int j = i; j > 0 && env_cost(handles[order[j]]->env) > env_cost(handles[order[j - 1]]->env); --j) {
            const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp;
        }
    MixedArgs m;
    memset(&m, 0, sizeof m);
    m.n_seg = n_handles;
    uint32_t blocks = 0;
    for (int i = 0; i < n_handles; ++i) {
        nig_handle *h = handles[order[i]];
        const int64_t o = lane_offsets[order[i]];
        RolloutArgs &q = m.seg[i];
        q.s = base_step_args(h);
        q.s.actions = action_ring + o; q.s.ld_act = (uint32_t)ld_act;
        q.s.reward = reward_out ? reward_out + o : nullptr;
        q.s.flags = flags_out ? flags_out + o : nullptr;
        q.s.t_ptr = nullptr; q.s.t_off = h->t;
        q.n_steps = n_steps; q.it0 = 0; q.ring_len = ring_len; q.slot_stride = (uint32_t)slot_stride;
        q.out_stride = (uint32_t)out_stride; q.block0 = 0;
        if (obs_out) { q.obs_out = obs_out + o; q.ld_obs_out = (uint32_t)ld_obs; q.obs_step_stride = (uint64_t)obs_step_stride; q.obs_aos = 0; }
        blocks += grid_for(h->B);
        m.blk_end[i] = blocks;
        m.env[i] = h->env;
    }
    HIP_TRY(hipSetDevice(handles[0]->device));
    nig_launch_mixed_rollout(!reward_out ? 0 : (obs_out ? 2 : 1), m, blocks, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    for (int k = 0; k < n_handles; ++k) handles[k]->t += (uint32_t)n_steps;
    return NIG_OK;
}

int nig_create_mixed(int32_t n_segments, const int32_t *env_ids, const int64_t *counts, int device, uint64_t seed,
                     uint64_t env_index0, uint32_t flags, nig_mixed **out)
{
    if (!out) return fail(NIG_ERR_INVALID, "nig_create_mixed: out is NULL%s");
    *out = nullptr;
    if (!env_ids || !counts || n_segments <= 0 || n_segments > NIG_MIXED_MAX_SEGMENTS)
        return fail(NIG_ERR_INVALID, "nig_create_mixed: 1..12 segments%s");
    nig_mixed *m = new (std::nothrow) nig_mixed();
    if (!m) return fail(NIG_ERR_INVALID, "nig_create_mixed: out of host memory%s");
    memset(m, 0, sizeof *m);
    m->n = n_segments; m->device = device;
    int64_t off = 0;
    for (int k = 0; k < n_segments; ++k) {
        if (env_ids[k] < 0 || env_ids[k] >= NIG_NUM_ENVS || counts[k] <= 0 || counts[k] > NIG_MAX_BATCH) {
            delete m;
            return fail(NIG_ERR_INVALID, "nig_create_mixed: bad env id or lane count%s");
        }
        m->off[k] = off;
        off += align_up(counts[k], BLOCK);            // every segment starts on a block boundary
        m->lanes += counts[k];
        if (SPECS[env_ids[k]].state_dim > m->s_max) m->s_max = SPECS[env_ids[k]].state_dim;
        if (SPECS[env_ids[k]].action_dim > m->a_max) m->a_max = SPECS[env_ids[k]].action_dim;
    }
    m->ld = off;
    if (m->ld > NIG_MAX_PITCH) { delete m; return fail(NIG_ERR_INVALID, "nig_create_mixed: more than 2^26 columns%s"); }
    int rc = NIG_OK;
    for (int k = 0; k < n_segments && rc == NIG_OK; ++k)
        rc = nig_create(env_ids[k], counts[k], device, seed, env_index0 + (uint64_t)m->off[k], 0, 0.0, flags, nullptr, &m->seg[k]);
    if (rc == NIG_OK) {
        const size_t bytes = (size_t)m->s_max * m->ld * sizeof(float);
        hipError_t e = hipMalloc((void **)&m->state, bytes);
        if (e == hipSuccess) e = hipMemset(m->state, 0, bytes);      // rows >= S of a segment stay zero
        if (e != hipSuccess) rc = fail(NIG_ERR_HIP, "nig_create_mixed: state matrix: %s", hipGetErrorString(e));
    }
    for (int k = 0; k < n_segments && rc == NIG_OK; ++k) rc = nig_bind_state(m->seg[k], m->state + m->off[k], m->ld);
    if (rc != NIG_OK) { nig_mixed_destroy(m); return rc; }
    *out = m;
    return NIG_OK;
}

int nig_mixed_destroy(nig_mixed *m)
{
    if (!m) return NIG_OK;
    for (int k = 0; k < m->n; ++k) nig_destroy(m->seg[k]);
    if (m->state) (void)hipFree(m->state);
    delete m;
    return NIG_OK;
}

int nig_mixed_get_info(const nig_mixed *m, nig_mixed_info *out)
{
    if (!m || !out) return fail(NIG_ERR_INVALID, "nig_mixed_get_info: NULL argument%s");
    memset(out, 0, sizeof *out);
    out->n_segments = m->n; out->state_dim_max = m->s_max; out->action_dim_max = m->a_max;
    out->lanes = m->lanes; out->ld = m->ld;
    for (int k = 0; k < m->n; ++k) { out->env[k] = m->seg[k]->env; out->offset[k] = m->off[k]; out->count[k] = m->seg[k]->B; }
    return NIG_OK;
}

float *nig_mixed_state(const nig_mixed *m) { return m ? m->state : nullptr; }

nig_handle *nig_mixed_segment(const nig_mixed *m, int32_t k) { return (m && k >= 0 && k < m->n) ? m->seg[k] : nullptr; }

int nig_mixed_reset(nig_mixed *m, void *stream)
{
    if (!m) return fail(NIG_ERR_INVALID, "nig_mixed_reset: NULL handle%s");
    for (int k = 0; k < m->n; ++k) {
        const int rc = nig_reset(m->seg[k], nullptr, nullptr, 0, stream);
        if (rc != NIG_OK) return rc;
    }
    return NIG_OK;
}

int nig_mixed_fill_actions(nig_mixed *m, uint32_t t, float *actions, void *stream)
{
    if (!m || !actions) return fail(NIG_ERR_INVALID, "nig_mixed_fill_actions: NULL argument%s");
    for (int k = 0; k < m->n; ++k) {
        const int rc = nig_fill_actions(m->seg[k], t, actions + m->off[k], m->ld, stream);
        if (rc != NIG_OK) return rc;
    }
    return NIG_OK;
}

int nig_mixed_step(nig_mixed *m, const float *actions, float *reward_out, uint32_t *flags_out, void *stream)
{
    if (!m || !actions) return fail(NIG_ERR_INVALID, "nig_mixed_step: NULL argument%s");
    for (int k = 0; k < m->n; ++k) {              // one step kernel per segment (each its own env type), same stream
        const int64_t o = m->off[k];
        const int rc = nig_step(m->seg[k], actions + o, m->ld, nullptr, nullptr, 0, reward_out ? reward_out + o : nullptr, nullptr,
                                flags_out ? flags_out + o : nullptr, nullptr, 0, stream);
        if (rc != NIG_OK) return rc;
    }
    return NIG_OK;
}

int nig_mixed_rollout(nig_mixed *m, int32_t n_steps, const float *action_ring, int64_t slot_stride, int32_t ring_len,
                      float *reward_out, uint32_t *flags_out, int64_t out_stride, void *stream)
{
    if (!m) return fail(NIG_ERR_INVALID, "nig_mixed_rollout: NULL handle%s");
    return nig_rollout_mixed(m->seg, m->off, m->n, n_steps, action_ring, m->ld, slot_stride, ring_len, reward_out, flags_out,
                             out_stride, stream);
}

int nig_mixed_rollout_obs(nig_mixed *m, int32_t n_steps, const float *action_ring, int64_t slot_stride, int32_t ring_len,
                          float *reward_out, uint32_t *flags_out, int64_t out_stride, float *obs_out, int64_t obs_step_stride,
                          void *stream)
{
    if (!m) return fail(NIG_ERR_INVALID, "nig_mixed_rollout_obs: NULL handle%s");
    return nig_rollout_mixed_obs(m->seg, m->off, m->n, n_steps, action_ring, m->ld, slot_stride, ring_len, reward_out, flags_out,
                                 out_stride, obs_out, m->ld, obs_step_stride, stream);
}

// ---- the path's one collective: all-gather of the partial tallies over RCCL ----------------------
// RCCL is resolved at run time from whatever copy the process already holds (a PyTorch process has its own
// librccl.so; a plain C host links or loads /opt/rocm/lib/librccl.so): a communicator is only meaningful to the
// library instance that created it, so libnig.so must not bring a second one.
typedef int (*nccl_allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*nccl_count_fn)(void *, int *);
static bool rccl_resolve(nccl_allgather_fn *ag, nccl_count_fn *cnt)
{
    static nccl_allgather_fn s_ag = nullptr;
    static nccl_count_fn s_cnt = nullptr;
    if (!s_ag) {
        void *f = dlsym(RTLD_DEFAULT, "ncclAllGather"), *c = dlsym(RTLD_DEFAULT, "ncclCommCount");
        if (!f || !c) {
            void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (lib) { f = dlsym(lib, "ncclAllGather"); c = dlsym(lib, "ncclCommCount"); }
        }
        s_ag = (nccl_allgather_fn)f; s_cnt = (nccl_count_fn)c;
    }
    *ag = s_ag; *cnt = s_cnt;
    return s_ag && s_cnt;
}

int nig_reduce_metrics(nig_handle *const *handles, int32_t n_handles, void *comm, double *scratch, int64_t scratch_doubles,
                       double *out, void *stream)
{
    if (!handles || n_handles <= 0 || !scratch || !out) return fail(NIG_ERR_INVALID, "nig_reduce_metrics: NULL argument%s");
    int world = 1;
    nccl_allgather_fn allgather = nullptr;
    nccl_count_fn count = nullptr;
    if (comm) {
        if (!rccl_resolve(&allgather, &count)) return fail(NIG_ERR_UNSUPPORTED, "nig_reduce_metrics: RCCL (librccl.so) not found in this process%s");
        if (count(comm, &world) != 0 || world <= 0) return fail(NIG_ERR_INVALID, "nig_reduce_metrics: ncclCommCount failed%s");
    }
    // scratch: [n_handles][ROWS] local partials | [ROWS] this rank's partial | [world][ROWS] gathered
    const int64_t need = ((int64_t)n_handles + 1 + world) * NIG_T_ROWS;
    if (scratch_doubles < need) return fail(NIG_ERR_INVALID, "nig_reduce_metrics: scratch smaller than (n_handles + 1 + ranks) * NIG_T_ROWS doubles%s");
    hipStream_t st = (hipStream_t)stream;
    double *local = scratch, *mine = scratch + (int64_t)n_handles * NIG_T_ROWS, *all = mine + NIG_T_ROWS;
    for (int k = 0; k < n_handles; ++k) {
        const int rc = nig_reduce_tally(handles[k], local + (int64_t)k * NIG_T_ROWS, stream);
        if (rc != NIG_OK) return rc;
    }
    hipLaunchKernelGGL(combine_partials_kernel, dim3(1), dim3(64), 0, st, (const double *)local, n_handles, mine);
    if (comm) {
        // ncclAllGather (RCCL; over xGMI between the GPUs of a node): NIG_T_ROWS doubles per rank, rank order
        if (allgather(mine, all, (size_t)NIG_T_ROWS, /*ncclDouble*/ 8, comm, st) != 0)
            return fail(NIG_ERR_HIP, "nig_reduce_metrics: ncclAllGather failed%s");
        hipLaunchKernelGGL(combine_partials_kernel, dim3(1), dim3(64), 0, st, (const double *)all, world, out);
    } else {
        hipLaunchKernelGGL(combine_partials_kernel, dim3(1), dim3(64), 0, st, (const double *)mine, 1, out);
    }
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

int nig_fill_actions(nig_handle *h, uint32_t t, float *actions, int64_t ld_act, void *stream)
{
    if (!h || !actions || ld_act < h->B) return fail(NIG_ERR_INVALID, "nig_fill_actions: bad argument%s");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t lo = (uint32_t)h->seed, hi = (uint32_t)(h->seed >> 32);
    launch_of(h->env)->fill(actions, ld_act, h->B, h->env0, lo, hi, t, grid_for(h->B), st);
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

int nig_set_state(nig_handle *h, const float *state, int64_t ld, const uint32_t *ctr, void *stream)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_set_state: NULL handle%s");
    hipStream_t st = (hipStream_t)stream;
    const nig_layout &L = h->lay;
    if (state) {
        if (ld < h->B) return fail(NIG_ERR_INVALID, "nig_set_state: ld < batch%s");
        hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(h->B)), dim3(BLOCK), 0, st, state, ld,
                           h->state, h->ld_state, SPECS[h->env].state_dim, h->B);
        HIP_TRY(hipGetLastError());
    }
    if (ctr) {
        HIP_TRY(hipMemcpyAsync(h->ws + L.off_ctr, ctr, (size_t)h->B * 4, hipMemcpyDeviceToDevice, st));
        h->may_hold_done = true;                  // the caller's counter words may mark lanes done
    }
    return NIG_OK;
}

int nig_get_state(nig_handle *h, float *state, int64_t ld, uint32_t *ctr, void *stream)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_get_state: NULL handle%s");
    hipStream_t st = (hipStream_t)stream;
    const nig_layout &L = h->lay;
    if (state) {
        if (ld < h->B) return fail(NIG_ERR_INVALID, "nig_get_state: ld < batch%s");
        hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(h->B)), dim3(BLOCK), 0, st,
                           (const float *)h->state, h->ld_state, state, ld, SPECS[h->env].state_dim, h->B);
        HIP_TRY(hipGetLastError());
    }
    if (ctr) HIP_TRY(hipMemcpyAsync(ctr, h->ws + L.off_ctr, (size_t)h->B * 4, hipMemcpyDeviceToDevice, st));
    return NIG_OK;
}

int nig_get_safety_metrics(nig_handle *h, const uint32_t *flags, int32_t *out, int64_t ld_out, void *stream)
{
    if (!h || !flags || !out || ld_out < h->B) return fail(NIG_ERR_INVALID, "nig_get_safety_metrics: bad argument%s");
    hipLaunchKernelGGL(safety_metrics_kernel, dim3(grid_for(h->B)), dim3(BLOCK), 0, (hipStream_t)stream, flags, out, ld_out, h->B,
                       enabled_constraints(h));
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

int nig_reduce_tally(nig_handle *h, double *partial_out, void *stream)
{
    if (!h || !partial_out) return fail(NIG_ERR_INVALID, "nig_reduce_tally: NULL argument%s");
    if (h->lay.off_tally < 0) return fail(NIG_ERR_INVALID, "nig_reduce_tally: handle created without NIG_F_TALLY%s");
    hipStream_t st = (hipStream_t)stream;
    int nblk = (int)((h->B + BLOCK - 1) / BLOCK);
    if (nblk > REDUCE_BLOCKS) nblk = REDUCE_BLOCKS;
    hipLaunchKernelGGL(reduce_tally_stage1, dim3(nblk), dim3(BLOCK), 0, st, (const double *)(h->ws + h->lay.off_tally),
                       h->lay.ld, h->B, h->scratch);
    hipLaunchKernelGGL(reduce_tally_stage2, dim3(1), dim3(REDUCE_BLOCKS), 0, st, (const double *)h->scratch, nblk, partial_out);
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

}  // extern "C"
