// nig_kernels.hip -- HIP kernels (gfx950) and C-ABI of libnig.so.
//
// One wavefront lane per environment instance.  State, actions, noise and outputs are
// structure-of-arrays ([row][lane], row pitch ld) so every global access of a wave is one
// fully coalesced 256-byte row segment.  The step kernel fuses the whole of
// IndustrialEnv.step (environments/base.py:157-213): clip -> constraint checks on the
// pre-state -> dynamics -> reward -> penalties -> counters -> done/truncation -> critical
// shutdown -> (optional) episode tally flush and in-kernel auto-reset.  No MFMA: these are
// elementwise ODE updates (HBM-bound, DESIGN.md "Roofline").
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see __graft_entry__.build).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "../../include/nig.h"
#include "nig_envs.hpp"

namespace nig {

constexpr int BLOCK = 256;
constexpr int REDUCE_BLOCKS = 256;

struct StepArgs {
    // library-owned
    float *state; uint32_t *ctr; long long *life_viol; double *ep_ret; double *tally;
    int64_t ld; int64_t B;
    // caller-owned
    const float *actions; int64_t ld_act;
    const double *step_noise; const double *reset_noise; int64_t ld_noise;
    float *reward; double *reward64; uint32_t *flags; float *final_obs; int64_t ld_obs;
    // scalars
    uint64_t env0; uint32_t seed_lo, seed_hi;
    const uint32_t *t_ptr; uint32_t t_off;   // launch counter t = (t_ptr ? *t_ptr : 0) + t_off (graph replay keeps t on the device)
    int max_steps; float dt32; double dt; uint32_t hflags; uint32_t cmask;
};

template <class Env>
struct StepResult {
    typename Env::reward_t reward;
    uint32_t viol_bits;
    int nviol, ncrit;
    bool terminated, truncated;
};

// IndustrialEnv.step for one lane, entirely in registers (base.py:157-213).
template <class Env>
__device__ __forceinline__ void step_core(const float (&s)[Env::S], float (&a)[Env::A],
                                          const double (&nz)[Env::KS > 0 ? Env::KS : 1], int step_pre,
                                          int max_steps, float dt32, double dt, uint32_t cmask,
                                          float (&n)[Env::S], StepResult<Env> &out)
{
    using R = typename Env::reward_t;
#pragma unroll
    for (int k = 0; k < Env::A; ++k) {            // base.py:167 np.clip(action, -1, 1) == min(max(x,lo),hi)
        float x = a[k];
        x = (x < -1.0f) ? -1.0f : x;
        x = (x > 1.0f) ? 1.0f : x;
        a[k] = x;
    }
    const uint32_t vb = Env::violated(s, a) & cmask;   // base.py:170 (and again :180, same inputs); cmask: base.py:224-228
    Env::dynamics(s, a, nz, dt32, dt, n);         // base.py:173
    R r = Env::reward(n, a);                      // base.py:176
#pragma unroll
    for (int k = 0; k < 3; ++k)                   // base.py:179-183, constraint order
        r = (vb & (1u << k)) ? (R)(r + (R)Env::penalty(k)) : r;
    const int nviol = __popc(vb);
    const int ncrit = __popc(vb & Env::CRIT_MASK);
    bool term = Env::done(n);                     // base.py:190
    const bool trunc = (step_pre + 1) >= max_steps;   // base.py:191
    if (ncrit > 0) { term = true; r = r - (R)1000; }  // base.py:195-198
    out.reward = r; out.viol_bits = vb; out.nviol = nviol; out.ncrit = ncrit;
    out.terminated = term; out.truncated = trunc;
}

template <class Env, bool PARITY>
__global__ void __launch_bounds__(BLOCK) step_kernel(const StepArgs p)
{
    constexpr int S = Env::S, A = Env::A, KS = Env::KS, KR = Env::KR;
    constexpr int KSN = KS > 0 ? KS : 1;
    using R = typename Env::reward_t;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= p.B) return;

    const uint32_t ctr = p.ctr[i];
    if (ctr & NIG_CTR_DONE) {                     // base.py:159-160: finished, waiting for reset
        if (p.flags) p.flags[i] = NIG_FLAG_INACTIVE | ((ctr & NIG_CTR_STEP_MASK) << NIG_FLAG_STEP_SHIFT);
        if (p.reward) p.reward[i] = 0.0f;
        if (p.reward64) p.reward64[i] = 0.0;
        return;
    }
    float s[S], a[A], n[S];
    double nz[KSN];
#pragma unroll
    for (int k = 0; k < S; ++k) s[k] = p.state[(int64_t)k * p.ld + i];
#pragma unroll
    for (int k = 0; k < A; ++k) a[k] = p.actions[(int64_t)k * p.ld_act + i];

    RngKey key;
    {
        const uint64_t gi = p.env0 + (uint64_t)i;
        key.env_lo = (uint32_t)gi; key.env_hi = (uint32_t)(gi >> 32);
        key.t = (p.t_ptr ? *p.t_ptr : 0u) + p.t_off; key.seed_lo = p.seed_lo; key.seed_hi = p.seed_hi;
    }
    if constexpr (KS > 0) {
        if constexpr (PARITY) {
#pragma unroll
            for (int k = 0; k < KS; ++k) nz[k] = p.step_noise[(int64_t)k * p.ld_noise + i];
        } else {
            Env::draw_step(key, nz);
        }
    } else {
        nz[0] = 0.0;
    }

    const int step_pre = (int)(ctr & NIG_CTR_STEP_MASK);
    StepResult<Env> res;
    step_core<Env>(s, a, nz, step_pre, p.max_steps, p.dt32, p.dt, p.cmask, n, res);

    const int step = step_pre + 1;
    const uint32_t viol_ep = (ctr >> NIG_CTR_VIOL_SHIFT) + (uint32_t)res.nviol;   // base.py:182
    const bool done = res.terminated || res.truncated;
    uint32_t fl = (res.terminated ? NIG_FLAG_TERMINATED : 0u) | (res.truncated ? NIG_FLAG_TRUNCATED : 0u) |
                  (res.viol_bits << NIG_FLAG_VIOL_SHIFT) | ((uint32_t)res.nviol << NIG_FLAG_NVIOL_SHIFT) |
                  ((uint32_t)res.ncrit << NIG_FLAG_NCRIT_SHIFT) | (res.ncrit > 0 ? NIG_FLAG_SHUTDOWN : 0u) |
                  ((uint32_t)step << NIG_FLAG_STEP_SHIFT);
    uint32_t nctr = (uint32_t)step | (viol_ep << NIG_CTR_VIOL_SHIFT);

    double ret = 0.0;
    if (p.tally) {                                // utils.py:99  episode_return += reward
        if constexpr (sizeof(R) == 4) ret = (double)((float)p.ep_ret[i] + res.reward);   // float32 accumulation (CR)
        else ret = p.ep_ret[i] + (double)res.reward;
    }
    if (done) {
        p.life_viol[i] += (long long)viol_ep;     // base.py:183 total_violations (never reset, base.py:139-141)
        if (p.tally) {                            // utils.py:120-125 per-episode bookkeeping
            double *T = p.tally + i;
            const int64_t ld = p.ld;
            const double len = (double)step;
            T[NIG_T_EPISODES * ld] += 1.0;
            T[NIG_T_RET_SUM * ld] += ret;
            T[NIG_T_RET_SQ * ld] += ret * ret;
            T[NIG_T_RET_MIN * ld] = fmin(T[NIG_T_RET_MIN * ld], ret);
            T[NIG_T_RET_MAX * ld] = fmax(T[NIG_T_RET_MAX * ld], ret);
            T[NIG_T_LEN_SUM * ld] += len;
            T[NIG_T_LEN_SQ * ld] += len * len;
            T[NIG_T_VIOL * ld] += (double)viol_ep;
            T[NIG_T_CRIT * ld] += (double)res.ncrit;   // a critical step always ends the episode
            T[NIG_T_SHUTDOWN * ld] += (res.ncrit > 0) ? 1.0 : 0.0;
            T[NIG_T_SUCCESS * ld] += (ret > 0.0) ? 1.0 : 0.0;
            ret = 0.0;
        }
        if (p.final_obs) {
#pragma unroll
            for (int k = 0; k < S; ++k) p.final_obs[(int64_t)k * p.ld_obs + i] = n[k];
        }
        if (p.hflags & NIG_F_AUTORESET) {         // base.py:133-155 for this lane, same launch
            double rn[KR];
            if constexpr (PARITY) {
#pragma unroll
                for (int k = 0; k < KR; ++k) rn[k] = p.reset_noise[(int64_t)k * p.ld_noise + i];
            } else {
                Env::draw_init(key, rn);
            }
            Env::init(rn, n);
            nctr = 0u;
            fl |= NIG_FLAG_DID_RESET;
        } else {
            nctr |= NIG_CTR_DONE;
        }
    }
#pragma unroll
    for (int k = 0; k < S; ++k) p.state[(int64_t)k * p.ld + i] = n[k];
    p.ctr[i] = nctr;
    if (p.tally) p.ep_ret[i] = ret;
    if (p.reward) p.reward[i] = (float)res.reward;
    if (p.reward64) p.reward64[i] = (double)res.reward;
    if (p.flags) p.flags[i] = fl;
}

struct ResetArgs {
    float *state; uint32_t *ctr; long long *life_viol; double *ep_ret;
    int64_t ld; int64_t B;
    const uint8_t *mask; const double *noise; int64_t ld_noise;
    uint64_t env0; uint32_t seed_lo, seed_hi, t;
};

template <class Env, bool PARITY>
__global__ void __launch_bounds__(BLOCK) reset_kernel(const ResetArgs p)
{
    constexpr int S = Env::S, KR = Env::KR;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= p.B) return;
    if (p.mask && !p.mask[i]) return;
    double rn[KR];
    if constexpr (PARITY) {
#pragma unroll
        for (int k = 0; k < KR; ++k) rn[k] = p.noise[(int64_t)k * p.ld_noise + i];
    } else {
        RngKey key;
        const uint64_t gi = p.env0 + (uint64_t)i;
        key.env_lo = (uint32_t)gi; key.env_hi = (uint32_t)(gi >> 32);
        key.t = p.t; key.seed_lo = p.seed_lo; key.seed_hi = p.seed_hi;
        Env::draw_init(key, rn);
    }
    float s[S];
    Env::init(rn, s);
#pragma unroll
    for (int k = 0; k < S; ++k) p.state[(int64_t)k * p.ld + i] = s[k];
    const uint32_t ctr = p.ctr[i];
    // violations of an abandoned (not finished) episode still belong to total_violations
    if (!(ctr & NIG_CTR_DONE)) p.life_viol[i] += (long long)(ctr >> NIG_CTR_VIOL_SHIFT);
    p.ctr[i] = 0u;                                // base.py:137-139
    if (p.ep_ret) p.ep_ret[i] = 0.0;
}

template <class Env>
__global__ void __launch_bounds__(BLOCK) fill_actions_kernel(float *act, int64_t ld_act, int64_t B, uint64_t env0,
                                                             uint32_t seed_lo, uint32_t seed_hi, uint32_t t)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= B) return;
    RngKey key;
    const uint64_t gi = env0 + (uint64_t)i;
    key.env_lo = (uint32_t)gi; key.env_hi = (uint32_t)(gi >> 32);
    key.t = t; key.seed_lo = seed_lo; key.seed_hi = seed_hi;
    double u[Env::A];
    gen_uniforms<Env::A>(key, STREAM_ACTION, u);
#pragma unroll
    for (int k = 0; k < Env::A; ++k) act[(int64_t)k * ld_act + i] = (float)(2.0 * u[k] - 1.0);
}

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

__global__ void __launch_bounds__(BLOCK) safety_metrics_kernel(const uint32_t *flags, int32_t *out, int64_t ld_out,
                                                               int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= B) return;
    const uint32_t f = flags[i];
    const int nv = (int)((f >> NIG_FLAG_NVIOL_SHIFT) & 3u), nc = (int)((f >> NIG_FLAG_NCRIT_SHIFT) & 3u);
    out[0 * ld_out + i] = 3 - nv;   // constraints_satisfied   base.py:96-106
    out[1 * ld_out + i] = 3;        // total_constraints       base.py:115
    out[2 * ld_out + i] = nv;       // violation_count
    out[3 * ld_out + i] = nc;       // critical_violations
    out[4 * ld_out + i] = 3 - nv;   // safety_score * total    base.py:116
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

__global__ void reduce_tally_stage2(const double *scratch, int nblk, double *out)
{
    const int r = threadIdx.x;
    if (r >= NIG_T_ROWS) return;
    const bool is_min = (r == NIG_T_RET_MIN), is_max = (r == NIG_T_RET_MAX);
    double acc = is_min ? __builtin_inf() : (is_max ? -__builtin_inf() : 0.0);
    for (int b = 0; b < nblk; ++b) {
        const double v = scratch[(int64_t)b * NIG_T_ROWS + r];
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
};
static const char *NAMES[NIG_NUM_ENVS] = {"ChemicalReactor-v0", "PowerGrid-v0", "RobotAssembly-v0"};

template <class Env>
static void launch_reset(const ResetArgs &a, bool parity, hipStream_t st)
{
    if (parity) hipLaunchKernelGGL((reset_kernel<Env, true>), dim3(grid_for(a.B)), dim3(BLOCK), 0, st, a);
    else hipLaunchKernelGGL((reset_kernel<Env, false>), dim3(grid_for(a.B)), dim3(BLOCK), 0, st, a);
}

template <class Env>
static void launch_step(const StepArgs &a, bool parity, hipStream_t st)
{
    if (parity) hipLaunchKernelGGL((step_kernel<Env, true>), dim3(grid_for(a.B)), dim3(BLOCK), 0, st, a);
    else hipLaunchKernelGGL((step_kernel<Env, false>), dim3(grid_for(a.B)), dim3(BLOCK), 0, st, a);
}

static StepArgs base_step_args(const nig_handle *h)
{
    const nig_layout &L = h->lay;
    StepArgs a;
    memset(&a, 0, sizeof a);
    a.state = (float *)(h->ws + L.off_state); a.ctr = (uint32_t *)(h->ws + L.off_ctr);
    a.life_viol = (long long *)(h->ws + L.off_life_viol);
    a.ep_ret = L.off_ep_return >= 0 ? (double *)(h->ws + L.off_ep_return) : nullptr;
    a.tally = L.off_tally >= 0 ? (double *)(h->ws + L.off_tally) : nullptr;
    a.ld = L.ld; a.B = h->B;
    a.env0 = h->env0; a.seed_lo = (uint32_t)h->seed; a.seed_hi = (uint32_t)(h->seed >> 32);
    a.max_steps = h->max_steps; a.dt32 = (float)h->dt; a.dt = h->dt; a.hflags = h->flags; a.cmask = h->cmask;
    return a;
}

static void dispatch_step(const nig_handle *h, const StepArgs &a, bool parity, hipStream_t st)
{
    switch (h->env) {
    case NIG_ENV_CHEMICAL_REACTOR: launch_step<ChemicalReactor>(a, parity, st); break;
    case NIG_ENV_POWER_GRID: launch_step<PowerGrid>(a, parity, st); break;
    default: launch_step<RobotAssembly>(a, parity, st); break;
    }
}

extern "C" {

const char *nig_version(void) { return "nig 0.1.0 (gfx950)"; }
const char *nig_last_error(void) { return g_err; }

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
    off = align_up(off + (int64_t)REDUCE_BLOCKS * NIG_T_ROWS * 8, 256) + 256;
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
    if (batch <= 0 || batch > ((int64_t)1 << 40)) return fail(NIG_ERR_INVALID, "nig_create: bad batch%s");
    if (max_episode_steps < 0 || max_episode_steps > NIG_MAX_EPISODE_STEPS)
        return fail(NIG_ERR_INVALID, "nig_create: max_episode_steps outside [1, 21845]%s");
    if (dt < 0.0 || dt != dt) return fail(NIG_ERR_INVALID, "nig_create: bad dt%s");
    if (env == NIG_ENV_CHEMICAL_REACTOR && dt != 0.0 && dt != 0.1)
        return fail(NIG_ERR_UNSUPPORTED, "nig_create: ChemicalReactor hard-codes dt=0.1 upstream (chemical_reactor.py:68)%s");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(NIG_ERR_NODEVICE, "nig_create: no HIP device (%s); there is no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= ndev) return fail(NIG_ERR_INVALID, "nig_create: device index out of range%s");
    HIP_TRY(hipSetDevice(device));

    nig_handle *h = new (std::nothrow) nig_handle();
    if (!h) return fail(NIG_ERR_INVALID, "nig_create: out of host memory%s");
    h->env = env; h->device = device; h->B = batch; h->seed = seed; h->env0 = env_index0;
    h->max_steps = max_episode_steps ? max_episode_steps : SPECS[env].max_episode_steps;
    h->dt = (dt != 0.0) ? dt : SPECS[env].dt;
    h->flags = flags; h->t = 0; h->cmask = 0x7u;
    nig_layout_query(env, batch, flags, &h->lay);
    if (workspace) {
        if (((uintptr_t)workspace & 255u) != 0) { delete h; return fail(NIG_ERR_INVALID, "nig_create: workspace not 256-byte aligned%s"); }
        h->ws = (char *)workspace; h->owns_ws = false;
    } else {
        void *p = nullptr;
        hipError_t me = hipMalloc(&p, (size_t)h->lay.bytes);
        if (me != hipSuccess) { delete h; return fail(NIG_ERR_HIP, "hipMalloc workspace: %s", hipGetErrorString(me)); }
        h->ws = (char *)p; h->owns_ws = true;
    }
    h->scratch = (double *)(h->ws + h->lay.bytes - 256 - align_up((int64_t)REDUCE_BLOCKS * NIG_T_ROWS * 8, 256));
    h->t_dev = (uint32_t *)(h->ws + h->lay.bytes - 256);
    const nig_layout &L = h->lay;
    hipLaunchKernelGGL(init_ws_kernel, dim3(grid_for(L.ld)), dim3(BLOCK), 0, (hipStream_t)0,
                       (uint32_t *)(h->ws + L.off_ctr), (long long *)(h->ws + L.off_life_viol),
                       L.off_ep_return >= 0 ? (double *)(h->ws + L.off_ep_return) : nullptr,
                       L.off_tally >= 0 ? (double *)(h->ws + L.off_tally) : nullptr, L.ld, L.batch);
    hipError_t le = hipGetLastError();
    if (le == hipSuccess) le = hipMemsetAsync(h->ws + L.off_state, 0, (size_t)SPECS[env].state_dim * L.ld * 4, (hipStream_t)0);
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

int nig_set_constraint_mask(nig_handle *h, uint32_t mask)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_set_constraint_mask: NULL handle%s");
    h->cmask = mask & 0x7u;
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
    a.state = (float *)(h->ws + L.off_state); a.ctr = (uint32_t *)(h->ws + L.off_ctr);
    a.life_viol = (long long *)(h->ws + L.off_life_viol);
    a.ep_ret = L.off_ep_return >= 0 ? (double *)(h->ws + L.off_ep_return) : nullptr;
    a.ld = L.ld; a.B = h->B; a.mask = mask; a.noise = init_noise; a.ld_noise = ld_noise;
    a.env0 = h->env0; a.seed_lo = (uint32_t)h->seed; a.seed_hi = (uint32_t)(h->seed >> 32); a.t = h->t;
    hipStream_t st = (hipStream_t)stream;
    switch (h->env) {
    case NIG_ENV_CHEMICAL_REACTOR: launch_reset<ChemicalReactor>(a, init_noise != nullptr, st); break;
    case NIG_ENV_POWER_GRID: launch_reset<PowerGrid>(a, init_noise != nullptr, st); break;
    default: launch_reset<RobotAssembly>(a, init_noise != nullptr, st); break;
    }
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

int nig_step(nig_handle *h, const float *actions, int64_t ld_act, const double *step_noise, const double *reset_noise,
             int64_t ld_noise, float *reward_out, double *reward64_out, uint32_t *flags_out, float *final_obs,
             int64_t ld_obs, void *stream)
{
    if (!h) return fail(NIG_ERR_INVALID, "nig_step: NULL handle%s");
    if (!actions || ld_act < h->B) return fail(NIG_ERR_INVALID, "nig_step: actions NULL or ld_act < batch%s");
    const nig_env_spec &sp = SPECS[h->env];
    const bool autoreset = (h->flags & NIG_F_AUTORESET) != 0;
    // parity mode = the caller supplies every value the reference's RNG would have drawn
    const bool parity = (step_noise != nullptr) || (reset_noise != nullptr);
    if (parity) {
        if (sp.k_step > 0 && !step_noise) return fail(NIG_ERR_INVALID, "nig_step: parity mode needs step_noise%s");
        if (autoreset && !reset_noise) return fail(NIG_ERR_INVALID, "nig_step: parity mode with auto-reset needs reset_noise%s");
        if (ld_noise < h->B) return fail(NIG_ERR_INVALID, "nig_step: ld_noise < batch%s");
    }
    if (final_obs && ld_obs < h->B) return fail(NIG_ERR_INVALID, "nig_step: ld_obs < batch%s");
    h->t += 1;
    StepArgs a = base_step_args(h);
    a.actions = actions; a.ld_act = ld_act;
    a.step_noise = step_noise; a.reset_noise = reset_noise; a.ld_noise = ld_noise;
    a.reward = reward_out; a.reward64 = reward64_out; a.flags = flags_out; a.final_obs = final_obs; a.ld_obs = ld_obs;
    a.t_ptr = nullptr; a.t_off = h->t;
    dispatch_step(h, a, parity, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

int nig_plan_create(nig_handle *h, int32_t n_steps, const float *action_ring, int64_t ld_act, int64_t slot_stride,
                    int32_t ring_len, float *reward_out, uint32_t *flags_out, int64_t out_stride, nig_plan **out)
{
    if (!out) return fail(NIG_ERR_INVALID, "nig_plan_create: out is NULL%s");
    *out = nullptr;
    if (!h || !action_ring || n_steps <= 0 || ring_len <= 0 || ld_act < h->B)
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
            a.actions = action_ring + (int64_t)slot * slot_stride; a.ld_act = ld_act;
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

int nig_fill_actions(nig_handle *h, uint32_t t, float *actions, int64_t ld_act, void *stream)
{
    if (!h || !actions || ld_act < h->B) return fail(NIG_ERR_INVALID, "nig_fill_actions: bad argument%s");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t lo = (uint32_t)h->seed, hi = (uint32_t)(h->seed >> 32);
    switch (h->env) {
    case NIG_ENV_CHEMICAL_REACTOR:
        hipLaunchKernelGGL((fill_actions_kernel<ChemicalReactor>), dim3(grid_for(h->B)), dim3(BLOCK), 0, st, actions, ld_act, h->B, h->env0, lo, hi, t);
        break;
    case NIG_ENV_POWER_GRID:
        hipLaunchKernelGGL((fill_actions_kernel<PowerGrid>), dim3(grid_for(h->B)), dim3(BLOCK), 0, st, actions, ld_act, h->B, h->env0, lo, hi, t);
        break;
    default:
        hipLaunchKernelGGL((fill_actions_kernel<RobotAssembly>), dim3(grid_for(h->B)), dim3(BLOCK), 0, st, actions, ld_act, h->B, h->env0, lo, hi, t);
        break;
    }
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
                           (float *)(h->ws + L.off_state), L.ld, SPECS[h->env].state_dim, h->B);
        HIP_TRY(hipGetLastError());
    }
    if (ctr) HIP_TRY(hipMemcpyAsync(h->ws + L.off_ctr, ctr, (size_t)h->B * 4, hipMemcpyDeviceToDevice, st));
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
                           (const float *)(h->ws + L.off_state), L.ld, state, ld, SPECS[h->env].state_dim, h->B);
        HIP_TRY(hipGetLastError());
    }
    if (ctr) HIP_TRY(hipMemcpyAsync(ctr, h->ws + L.off_ctr, (size_t)h->B * 4, hipMemcpyDeviceToDevice, st));
    return NIG_OK;
}

int nig_get_safety_metrics(nig_handle *h, const uint32_t *flags, int32_t *out, int64_t ld_out, void *stream)
{
    if (!h || !flags || !out || ld_out < h->B) return fail(NIG_ERR_INVALID, "nig_get_safety_metrics: bad argument%s");
    hipLaunchKernelGGL(safety_metrics_kernel, dim3(grid_for(h->B)), dim3(BLOCK), 0, (hipStream_t)stream, flags, out, ld_out, h->B);
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
    hipLaunchKernelGGL(reduce_tally_stage2, dim3(1), dim3(64), 0, st, (const double *)h->scratch, nblk, partial_out);
    HIP_TRY(hipGetLastError());
    return NIG_OK;
}

}  // extern "C"
