/*
 * nig.h -- C ABI of libnig.so: the MI355X-native batched simulator for the
 * IndustrialEnv.step() hot path of neoRL-industrial-gym.
 *
 * The reference has NO native/FFI layer: its boundary is the
 * Python class surface IndustrialEnv.reset/step (environments/base.py:133-213),
 * utils.make (utils.py:12-39) and utils.evaluate_with_safety (utils.py:42-154).
 * Each entry point below names the reference interface it stands in for.  The
 * Python mirror of that surface (neorl-industrial-gym_amd/) is a thin ctypes shim
 * over exactly these symbols; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns an int status, 0 = NIG_OK; nig_last_error() explains
 *     the last failure on the calling thread;
 *   - all data pointers are DEVICE pointers unless a parameter says "host";
 *   - batch arrays are structure-of-arrays: row k of a [R][B] array starts at
 *     element k*ld, where ld = nig_layout.ld (B rounded up to 64) for library-owned
 *     arrays and ld = the ld_* argument for caller-owned arrays;
 *   - every launch goes to the caller-supplied hipStream_t (passed as void*), no
 *     hidden synchronisation; a handle is not thread-safe, distinct handles are;
 *   - there is no CPU fallback: without a HIP device nig_create fails.
 */
#ifndef NIG_H
#define NIG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NIG_OK 0
#define NIG_ERR_INVALID 1      /* bad argument                                   */
#define NIG_ERR_HIP 2          /* a HIP runtime call failed                      */
#define NIG_ERR_NODEVICE 3     /* no usable HIP device                           */
#define NIG_ERR_UNSUPPORTED 4

/* environment ids: the registry of utils.py:26-32 (working environments) */
#define NIG_ENV_CHEMICAL_REACTOR 0   /* 'ChemicalReactor-v0'  environments/chemical_reactor.py */
#define NIG_ENV_POWER_GRID 1         /* 'PowerGrid-v0'        environments/power_grid.py       */
#define NIG_ENV_ROBOT_ASSEMBLY 2     /* 'RobotAssembly-v0'    environments/robot_assembly.py   */
/* Candidate rows: registered upstream (utils.py:30-31) but NOT instantiable there;
 * restated from the source text, no reference output exists to pin them. */
#define NIG_ENV_ADV_CHEMICAL_REACTOR 3   /* 'AdvancedChemicalReactor-v0'  environments/advanced_chemical_reactor.py */
#define NIG_ENV_ADV_POWER_GRID 4         /* 'AdvancedPowerGrid-v0'        environments/advanced_power_grid.py       */
/* BUILD-SPECIFIED environments: the reference's README lists them (README.md:28-32: name, dims,
 * constraint names) and ships no implementation -- no reference output exists, parity is undefined.
 * Plants: neorl-industrial-gym_amd/spec_plants.py.  They complete BASELINE's "all 7 envs" mixed batch. */
#define NIG_ENV_HVAC_CONTROL 5       /* 'HVACControl-v0'      18 / 5  */
#define NIG_ENV_WATER_TREATMENT 6    /* 'WaterTreatment-v0'   15 / 4  */
#define NIG_ENV_STEEL_ANNEALING 7    /* 'SteelAnnealing-v0'   20 / 6  */
#define NIG_ENV_SUPPLY_CHAIN 8       /* 'SupplyChain-v0'      28 / 10 */
#define NIG_NUM_ENVS 9

/* nig_create flags */
#define NIG_F_AUTORESET 0x1u   /* finished lanes re-sample their initial state inside the step kernel   */
#define NIG_F_TALLY 0x2u       /* keep per-lane episode tallies (what evaluate_with_safety aggregates)   */

/* per-lane flag word written by nig_step (one uint32 per env instance per step) */
#define NIG_FLAG_TERMINATED 0x1u        /* base.py:190,195-197                               */
#define NIG_FLAG_TRUNCATED 0x2u         /* base.py:191                                       */
#define NIG_FLAG_VIOL0 0x4u             /* constraint k violated on the pre-state: bit 2+k   */
#define NIG_FLAG_VIOL_SHIFT 2
#define NIG_FLAG_NVIOL_SHIFT 5          /* bits 5-6: SafetyMetrics.violation_count (0..3)    */
#define NIG_FLAG_NCRIT_SHIFT 7          /* bits 7-8: SafetyMetrics.critical_violations       */
#define NIG_FLAG_SHUTDOWN 0x200u        /* info['critical_shutdown'], base.py:210            */
#define NIG_FLAG_DID_RESET 0x400u       /* lane auto-reset in this call (state = new episode)*/
#define NIG_FLAG_INACTIVE 0x800u        /* lane was already done (no auto-reset): untouched  */
#define NIG_FLAG_VIOL3 0x1000u          /* 4th safety condition violated (Advanced envs)     */
#define NIG_FLAG_NVIOL_HI 0x2000u       /* adds 4 to the violation count field (Advanced envs) */
#define NIG_FLAG_STEP_SHIFT 16          /* bits 16-31: current_step after this call          */

/* per-lane counter word kept by the library */
#define NIG_CTR_STEP_MASK 0x7fffu       /* current_step (base.py:187)                        */
#define NIG_CTR_DONE 0x8000u            /* self.done (base.py:192,197)                       */
#define NIG_CTR_VIOL_SHIFT 16           /* violation_count of the running episode (base.py:182) */
#define NIG_MAX_EPISODE_STEPS 21845     /* keeps 3*steps inside 16 bits                      */
#define NIG_MAX_BATCH (1 << 24)         /* lanes per handle: row offsets stay 32-bit scalars */
#define NIG_MAX_PITCH (1 << 26)         /* largest caller row pitch (elements)               */

/* rows of the tally array (NIG_F_TALLY), all stored as double, one column per lane.
 * Count rows (EPISODES, LEN_SUM, VIOL, CRIT, SHUTDOWN, SUCCESS, SATISFIED, CONSTRAINTS) hold integers and every add to them is
 * an integer-valued float64 add: EXACT while a sum stays below 2^53 = 9.0e15 -- per lane, per GPU after nig_reduce_tally, and
 * across ranks after the rank-order combine ("counts must be reduced as integers, never fp32": no float32 is ever
 * involved; at the headline's 6.7e9 violations per 0.6 s a single GPU would need ~9 days of continuous stepping to reach the
 * bound).  A host that needs more sums the partial vectors of shorter runs (fresh handles) in integers itself;
 * parallel.metrics_from_partial rounds the float64 sums to int, which is the identity below 2^53. */
enum {
    NIG_T_EPISODES = 0,   /* finished episodes                                   utils.py:120 */
    NIG_T_RET_SUM,        /* sum of episode returns                              utils.py:130 */
    NIG_T_RET_SQ,         /* sum of squared episode returns                      utils.py:131 */
    NIG_T_RET_MIN,        /*                                                     utils.py:132 */
    NIG_T_RET_MAX,        /*                                                     utils.py:133 */
    NIG_T_LEN_SUM,        /* sum of episode lengths (= steps in finished eps)    utils.py:136 */
    NIG_T_LEN_SQ,         /*                                                     utils.py:137 */
    NIG_T_VIOL,           /* sum of SafetyMetrics.violation_count                utils.py:140 */
    NIG_T_CRIT,           /* sum of SafetyMetrics.critical_violations            utils.py:142 */
    NIG_T_SHUTDOWN,       /* steps with info['critical_shutdown']                utils.py:143 */
    NIG_T_SUCCESS,        /* episodes with return > 0                            utils.py:150 */
    NIG_T_SATISFIED,      /* sum over steps of SafetyMetrics.constraints_satisfied  utils.py:109 (numerator)   */
    NIG_T_CONSTRAINTS,    /* sum over steps of SafetyMetrics.total_constraints      utils.py:109 (denominator):
                             constraint_satisfaction_rate (utils.py:144-147) = SATISFIED / CONSTRAINTS, 1.0 if 0 */
    NIG_T_ROWS
};

/* static description of an environment: base.py:22-72 plus each env's ctor */
typedef struct nig_env_spec {
    int32_t state_dim;          /* base.py:41  */
    int32_t action_dim;         /* base.py:42  */
    int32_t n_constraints;      /* len(self.safety_constraints)              */
    int32_t max_episode_steps;  /* default: CR 500 (chemical_reactor.py:66), PG/RA 1000 (base.py:27) */
    int32_t k_step;             /* RNG values drawn per step, reference call order  */
    int32_t k_reset;            /* RNG values drawn per reset                       */
    double dt;                  /* base.py:28 */
    double penalty[3];          /* SafetyConstraint.penalty  */
    int32_t critical[3];        /* SafetyConstraint.critical */
    int32_t reward_is_f32;      /* 1: reward accumulates in float32 (CR, NumPy>=2), 0: fp64 */
} nig_env_spec;

/* byte offsets of the library-owned arrays inside the handle's workspace */
typedef struct nig_layout {
    int64_t batch;          /* B                                              */
    int64_t ld;             /* row pitch in elements of every [R][B] array    */
    int64_t bytes;          /* total workspace size                           */
    int64_t off_state;      /* float  [S][ld]  current state == observation   */
    int64_t off_ctr;        /* uint32 [ld]     counter word (NIG_CTR_*)       */
    int64_t off_life_viol;  /* int64  [ld]     total_violations of FINISHED episodes (base.py:57,183) */
    int64_t off_ep_return;  /* double [ld]     return of the running episode (NIG_F_TALLY only, else -1) */
    int64_t off_tally;      /* double [NIG_T_ROWS][ld]  (NIG_F_TALLY only, else -1) */
} nig_layout;

typedef struct nig_handle nig_handle;

/* "nig <major.minor.patch> (gfx950; generator <id>)".  The generator id names the fast-mode random stream: trajectories of a
 * given (seed, lane, launch counter) are reproducible only under the same id ("nig-philox-v1" up to 0.1.0; "nig-philox-v2":
 * Philox4x32-7, float32 reset draws, 0.2.0; "nig-philox-v3", since 0.3.0: v2 with PowerGrid's eight reset load factors taken as
 * 16-bit uniforms from the low bytes of its reset normals' words -- only PowerGrid trajectories differ from v2). */
const char *nig_version(void);
const char *nig_last_error(void);

/* Process-wide tuning knobs.  Results never depend on them: every kernel form a knob selects between is
 * bit-identical (tests/test_gpu_split.py).  No counterpart upstream.
 *   NIG_TUNE_SPLIT_BLOCKS   256-lane blocks per ROUND of the three-wave form of nig_rollout / nig_rollout_policy
 *                           (csrc/nig_split.hpp: producer / integrator / recorder wave per 64 ChemicalReactor or
 *                           RobotAssembly lanes, one block resident per compute unit); default = the compute units of the HANDLE's
 *                           device, read at nig_create (256 on an MI355X; nig_tune_get reports it for the device of
 *                           the latest handle), 0 = never use that form.  An explicit value is process-wide and atomic.  nig_rollout
 *                           uses it for batches of at most one round and (ChemicalReactor) for larger ones whose
 *                           last round is at least 3/4 full WHEN the launch writes an observation trajectory (without one the
 *                           rounds lose to lanes filling the SIMDs, measured in round 5), nig_rollout_policy (ChemicalReactor)
 *                           likewise.  The same
 *                           "one wave per SIMD" threshold selects PowerGrid's paired rollout form (a producer wave per
 *                           stepping wave, csrc/nig_pg_lds.hpp) and the step kernel's helper waves (nig_step / nig_plan_*
 *                           on auto-reset handles of ChemicalReactor, PowerGrid, RobotAssembly).  The environment variable
 *                           NIG_SPLIT_BLOCKS sets the initial value.
 *   NIG_TUNE_WIDE_MIN_BLOCKS  smallest batch, in 512-lane blocks, that nig_rollout runs in the WIDE form (csrc/nig_kernels.hpp
 *                           rollout_wide_kernel: PowerGrid, 512-thread blocks at four waves per SIMD,
 *                           handles on which no lane can be frozen); default = 1.5 x the compute units of the handle's
 *                           device + 1 (the first batch that no longer fits one round of the 256-lane form at three
 *                           blocks per compute unit); smaller batches run the same LDS-resident
 *                           body in 256-lane blocks; 2^30 or more = never use that body (rollout_kernel only).  Environment variable NIG_WIDE_MIN_BLOCKS sets the initial value.
 * A value of -1 removes an explicit setting: every handle is back on its own device's default (tests restore with it).
 * nig_tune returns NIG_OK or NIG_ERR_INVALID (unknown key / value below -1); nig_tune_get returns the value in effect
 * for the device of the latest handle, or -1 for an unknown key. */
/* NIG_TUNE_DIAG_RING_FAULT: exists in the TEST-ONLY library variant built with -DNIG_RING_SPIN_LIMIT=<polls>
 * (profiles/mkvariant.sh; csrc/nig_ring.hpp) -- there the cooperating-wave kernels' ring waits are bounded and a
 * time-out makes nig_rollout / nig_rollout_policy return NIG_ERR_HIP naming the ring (they synchronise the stream in that
 * build); a non-zero value makes the producing waves stop posting after 7 steps, which is how tests/test_gpu_ring_limit.py
 * shows the error path.  The production library returns NIG_ERR_UNSUPPORTED for this key. */
enum { NIG_TUNE_SPLIT_BLOCKS = 0, NIG_TUNE_WIDE_MIN_BLOCKS = 1, NIG_TUNE_DIAG_RING_FAULT = 2 };
int nig_tune(int32_t key, int64_t value);
int64_t nig_tune_get(int32_t key);
/* the value in effect for THIS handle (its own device's default unless an explicit setting exists); -1: NULL handle / unknown key */
int64_t nig_handle_tune_get(const nig_handle *h, int32_t key);

/* Measurement aid (no counterpart upstream: performance_benchmark.py:106-133 times its loop with time.time()).  Enqueues a
 * small kernel on `stream` of the CURRENT device that stamps, per compute unit, the shader-clock counter and the constant
 * 100 MHz counter into the caller's DEVICE buffer of NIG_CLOCK_STAMP_WORDS uint64: slot = XCD (HW_REG_XCC_ID, 0-7) x 256 +
 * HW_REG_HW_ID bits 15:8 (se_id, sh_id, cu_id); out[2 slot] = s_memtime, out[2 slot + 1] = s_memrealtime; a compute unit no
 * block landed on keeps its old words (zero the buffer first).  Two stamps around a run of launches give the clock the chip
 * held over that run: (memtime1 - memtime0) / (memrealtime1 - memrealtime0) x 100 MHz for every slot stamped both times --
 * bench.py records the median next to every timed region so that box-to-box spread of a line can be attributed (VERDICT r04
 * next #6).  Never part of a timed kernel. */
#define NIG_CLOCK_STAMP_WORDS 4096
int nig_clock_stamp(void *stream, uint64_t *out);

/* utils.make registry lookup (utils.py:26-35): name -> id, or -1 */
int nig_env_id(const char *name);
const char *nig_env_name(int env);
int nig_env_spec_get(int env, nig_env_spec *out);

/* Workspace size/layout for a batch; `flags` as for nig_create. */
int nig_layout_query(int env, int64_t batch, uint32_t flags, nig_layout *out);

/*
 * Construct a batch of `batch` independent env instances of one type on `device`
 * (stands in for B calls of utils.make / IndustrialEnv.__init__, base.py:22-72).
 *   env_index0         global index of lane 0 (keys the counter-based RNG so results do
 *                      not depend on how a job is sharded over GPUs)
 *   max_episode_steps  0 = the env's default; CR ignores overrides upstream
 *                      (chemical_reactor.py:62-69 hard-codes 500 / 0.1), here it is honoured
 *   dt                 0 = default
 *   workspace          device memory of nig_layout.bytes (256-byte aligned) owned by the
 *                      caller, or NULL to let the library hipMalloc it.  It must be ordinary
 *                      device-local memory of `device` (hipMalloc / a torch CUDA tensor), not
 *                      host-pinned or fine-grained memory: with NIG_F_TALLY the step kernel
 *                      updates the episode tally with float64 hardware atomics
 * States are undefined until nig_reset.
 */
int nig_create(int env, int64_t batch, int device, uint64_t seed, uint64_t env_index0,
               int32_t max_episode_steps, double dt, uint32_t flags, void *workspace,
               nig_handle **out);
int nig_destroy(nig_handle *h);
int nig_get_layout(const nig_handle *h, nig_layout *out);
void *nig_workspace(const nig_handle *h);

/* RNG launch counter t ("nig-philox-v2"): nig_reset draws with the current
 * t, nig_step increments t first; every draw is a pure function of (seed, global lane index, t),
 * whatever the launch shape (ChemicalReactor's two step draws come from the Philox block with
 * counter word (t+1)/2: words 0-1 for odd t, 2-3 for even t).  Exposed so a caller can
 * checkpoint / replay. */
int nig_get_counter(const nig_handle *h, uint32_t *t);
int nig_set_counter(nig_handle *h, uint32_t t);

/*
 * Keep this handle's state rows in a caller-owned SoA array instead of the workspace: row k
 * of lane i lives at state[k*ld + i].  Lets several handles of different env types share one
 * padded observation matrix [S_max][total lanes] (BASELINE config "all envs mixed-batch,
 * heterogeneous state dims, padded SoA"): bind each handle at its column offset.  The current
 * contents are not copied; call before nig_reset.  state == NULL restores the internal array.
 */
int nig_bind_state(nig_handle *h, float *state, int64_t ld);

/* env.remove_safety_constraint(name) for a built-in constraint (base.py:224-228): bit k of
 * `mask` keeps constraint k enabled (default 0x7).  A disabled constraint is neither
 * counted nor penalised.  Constraints ADDED by the user (base.py:220-222) are arbitrary
 * Python callables and stay on the host side of the binding. */
int nig_set_constraint_mask(nig_handle *h, uint32_t mask);

/*
 * IndustrialEnv.reset (base.py:133-155) for the lanes selected by `mask`
 * (uint8 [B], NULL = all): current_step = 0, done = False, violation_count = 0,
 * state = _get_initial_state().
 *   init_noise   double [k_reset][ld_noise]: the values the reference's RNG calls
 *                would have returned, in call order ("parity mode"); NULL = draw
 *                them on device from the counter-based generator ("fast mode").
 */
int nig_reset(nig_handle *h, const uint8_t *mask, const double *init_noise, int64_t ld_noise,
              void *stream);

/*
 * IndustrialEnv.step (base.py:157-213) for every lane, one kernel launch.
 *   actions      float [A][ld_act]   raw actions (clipped to [-1,1] inside, base.py:167)
 *   step_noise   double [k_step][ld_noise] or NULL (fast mode)
 *   reset_noise  double [k_reset][ld_noise] or NULL: initial-state draws for lanes that
 *                auto-reset in this call (NIG_F_AUTORESET)
 *   reward_out   float [B]   reward (rounded to float32 for PG/RA)        may be NULL
 *   reward64_out double [B]  the reference's fp64 Python-float reward      may be NULL
 *   flags_out    uint32 [B]  NIG_FLAG_* word                               may be NULL
 *   final_obs    float [S][ld_obs] terminal observation of lanes that finished in this
 *                call (other lanes untouched)                              may be NULL
 * The new state/observation is the library-owned state array (nig_layout.off_state);
 * with NIG_F_AUTORESET a finished lane already holds the first observation of its next
 * episode.  Without it a finished lane is frozen until nig_reset (the reference raises
 * RuntimeError, base.py:159-160; here its flag word reads NIG_FLAG_INACTIVE).
 */
int nig_step(nig_handle *h, const float *actions, int64_t ld_act,
             const double *step_noise, const double *reset_noise, int64_t ld_noise,
             float *reward_out, double *reward64_out, uint32_t *flags_out,
             float *final_obs, int64_t ld_obs, void *stream);

/*
 * IndustrialEnv.step with FLOAT64 actions: double [A][ld_act], everything else as nig_step.
 * The reference's own callers hand np.float64 arrays to step() -- get_dataset (chemical_reactor.py:364-393,
 * power_grid.py:216-233, robot_assembly.py:266-290) and the baseline agents (benchmarks/baseline_agents.py:28-114)
 * -- and base.py:167 clips without casting, so under NumPy >= 2 every expression that touches an action element is
 * float64 and float64 spreads through what depends on it until a value is stored into the float32 state vector
 * (ChemicalReactor: temperature, pressure, flows, concentration, level and the reward; PowerGrid: generation,
 * frequency deviation, the action penalty; RobotAssembly: the joint update, the action penalty).  This entry point
 * follows that arithmetic for the three NumPy envs (pinned by the reference's outputs, tests/golden/<env>_g5.npz and
 * _g6.npz: state words bit-exact); the other envs convert the action to float32 on entry, as their own code would.
 * reward_out is the float64 reward rounded to float32; ChemicalReactor's reward is float64 here (np.float32 - np.float64).
 */
int nig_step64(nig_handle *h, const double *actions, int64_t ld_act,
               const double *step_noise, const double *reset_noise, int64_t ld_noise,
               float *reward_out, double *reward64_out, uint32_t *flags_out,
               float *final_obs, int64_t ld_obs, void *stream);

/*
 * A plan = n_steps consecutive nig_step launches in fast mode recorded once as a hipGraph
 * and replayed with one call (the per-step launch is otherwise host-bound at small batch).
 * Step k (0-based) of every replay reads its actions from ring slot k % ring_len:
 *   action_ring  float, slot s at action_ring + s*slot_stride, each slot [A][ld_act]
 *   reward_out / flags_out  optional; slot s at base + s*out_stride (out_stride 0 = every
 *                step overwrites the same [B] array)
 * Semantics per step are exactly nig_step's (same kernel); the launch counter advances by
 * n_steps per replay.  This is the shape of the reference's own measurement loop
 * (performance_benchmark.py:106-133: sample action -> step -> reset on done).
 */
typedef struct nig_plan nig_plan;
int nig_plan_create(nig_handle *h, int32_t n_steps, const float *action_ring, int64_t ld_act,
                    int64_t slot_stride, int32_t ring_len, float *reward_out, uint32_t *flags_out,
                    int64_t out_stride, nig_plan **out);
int nig_plan_launch(nig_plan *p, void *stream);
int nig_plan_destroy(nig_plan *p);

/*
 * Fused rollout: n_steps consecutive steps of every lane in ONE kernel launch, state held in
 * registers between steps (fast mode only).  Same per-step semantics, generator keys and
 * bookkeeping as n_steps calls of nig_step -- the resulting state, counters and tallies are
 * bit-identical -- but a lane touches memory per step only for its action and for the outputs
 * requested here:
 *   action_ring / ld_act / slot_stride / ring_len   as for nig_plan_create; or, with ld_act == 0, a ROW-MAJOR ring: slot s =
 *                [batch][A] at action_ring + s*slot_stride (slot_stride >= batch*A) -- the layout of a policy's batched
 *                output (agents/base.py:106-141 predict() returns [n, A]).  Read natively (two 16-byte loads per lane, 2 KiB
 *                contiguous per wave) where every kernel of the launch is PowerGrid's LDS-resident body (auto-reset handle,
 *                no held lanes, batch a multiple of 256, 16-byte aligned slots; not the small-batch launches without an
 *                observation trajectory, whose stepping waves keep the state in registers); every
 *                other launch reads rows, so the library first transposes the slots this call reads (the first
 *                min(ring_len, n_steps): step k takes slot k % ring_len) into a buffer the handle owns (that many times
 *                A*ld floats, grown on demand, on the caller's stream, EVERY call: prefer rows for a ring that is
 *                replayed).  Same values either way (tests/test_gpu_action_layout.py).
 *   reward_out, flags_out   optional (both or neither), row of step k at base + k*out_stride
 *                (0 = overwrite)
 *   obs_out      optional (needs reward_out/flags_out) float trajectory: observation returned by step k (the terminal one for
 *                a lane that finishes) at obs_out + k*obs_step_stride, laid out [S][ld_obs], or
 *                -- with ld_obs == 0 -- row-major [B][S] (the D4RL observations[N,S] layout; obs_out
 *                16-byte aligned and obs_step_stride a multiple of 4 floats, i.e. B*S % 4 == 0 for a
 *                dense trajectory: each wave's 64 rows leave as whole-line 16-byte streaming stores).
 *                obs_step_stride == 0: every step overwrites the same block (as out_stride == 0 does for the
 *                reward / flag rows) -- the caller keeps the observations the last step returned
 * Stands in for the step loops of the reference's harnesses: benchmark_environment_steps
 * (performance_benchmark.py:106-133) and the get_dataset episode loops
 * (chemical_reactor.py:364-405, power_grid.py:209-237, robot_assembly.py:259-296).
 */
int nig_rollout(nig_handle *h, int32_t n_steps, const float *action_ring, int64_t ld_act,
                int64_t slot_stride, int32_t ring_len, float *reward_out, uint32_t *flags_out,
                int64_t out_stride, float *obs_out, int64_t ld_obs, int64_t obs_step_stride,
                void *stream);

/*
 * nig_rollout with the reference's RECORDED random draws in place of the generator's -- parity mode of the fused
 * kernels: the same kernel forms nig_rollout selects for the handle and batch (three-wave, LDS-resident wide / paired,
 * one-wave; same thresholds, same launch shapes) run base.py:157-213 end to end on the values np.random returned
 * inside _dynamics (chemical_reactor.py:149,159; power_grid.py:136-144) and _get_initial_state
 * (chemical_reactor.py:89-107, power_grid.py:90-110, robot_assembly.py:113-137).  One row set per call step, as
 * nig_step's parity arguments:
 *   step_noise   double, row set of call step k at step_noise + k*step_noise_stride, laid out [k_step][ld_noise]
 *                (NULL for an env without step noise)
 *   reset_noise  double, row set of call step k at reset_noise + k*reset_noise_stride, laid out [k_reset][ld_noise]:
 *                the initial-state draws of a lane whose episode ends in call step k (auto-reset handles; a finishing
 *                lane restarts from _get_initial_state on them, evaluated per lane)
 * action_ring must hold every step of the call (ring_len >= n_steps).  Outputs as nig_rollout with the row-major
 * trajectory (obs_out float [n_steps][B][S], obs_step_stride 0 or >= S*B): reward_out, flags_out and obs_out are all
 * required.  ChemicalReactor, PowerGrid, RobotAssembly only (NIG_ERR_UNSUPPORTED otherwise): the envs upstream can run.
 * Exists for tests/test_gpu_noise_rollout.py (tests/golden/<env>_g3.npz through every form); the arithmetic outside
 * the replaced draws is the code nig_rollout runs.
 */
int nig_rollout_noise(nig_handle *h, int32_t n_steps, const float *action_ring, int64_t ld_act,
                      int64_t slot_stride, int32_t ring_len, const double *step_noise, int64_t step_noise_stride,
                      const double *reset_noise, int64_t reset_noise_stride, int64_t ld_noise,
                      float *reward_out, uint32_t *flags_out, int64_t out_stride,
                      float *obs_out, int64_t obs_step_stride, void *stream);

/*
 * Closed-loop rollouts with an on-device policy ("nig-policy-v1").  The policy
 * families are the reference's non-neural ones: the baseline agents Constant / "MPC"
 * (proportional) / PID / Random (benchmarks/baseline_agents.py:28-114) and the behaviour
 * policies of get_dataset (chemical_reactor.py:364-393, power_grid.py:216-233,
 * robot_assembly.py:266-290): an affine map of the observation, plus per-dimension Gaussian
 * and uniform noise, an epsilon-mixture with a uniform random action, and a clip.
 * All arithmetic is float32, terms in ascending (k, then j) order, zero columns skipped:
 *     u_j = b_j + sum_k Wt[k][j] * obs[k]                       (kind AFFINE)
 *     e_j = setpoint_j - obs[j]; I_j += e_j; u_j = kp*e_j + ki*I_j + kd*(e_j - eprev_j)   (kind PID)
 *     u_j += sigma_j * z_j + half_range_j * (2*v_j - 1)          z ~ N(0,1), v ~ U[0,1)
 *     if (w < p_uniform) u_j = uniform_range * (2*r_j - 1)       w, r ~ U[0,1)
 *     a_j = min(max(u_j, clip_lo), clip_hi)                      then IndustrialEnv.step clips to [-1,1]
 * Random draws come from generator stream "policy" at (global lane, launch counter t).
 */
#define NIG_POLICY_AFFINE 1
#define NIG_POLICY_PID 2
#define NIG_MAX_STATE_DIM 32
#define NIG_MAX_ACTION_DIM 10
typedef struct nig_policy {
    int32_t kind;
    uint32_t colmask;        /* bit k set: observation column k has a non-zero weight (host-computed) */
    float Wt[NIG_MAX_STATE_DIM][NIG_MAX_ACTION_DIM];   /* Wt[k][j] = weight of obs[k] in action j */
    float b[NIG_MAX_ACTION_DIM];
    float sigma[NIG_MAX_ACTION_DIM];
    float half_range[NIG_MAX_ACTION_DIM];
    float p_uniform, uniform_range;
    float clip_lo, clip_hi;
    float kp, ki, kd;
    float setpoint[NIG_MAX_ACTION_DIM];
} nig_policy;

/* Install the policy used by nig_rollout_policy (copied to device memory owned by the handle).
 * A PID policy owns per-lane controller memory (integral, previous error: float [2*A][ld], device
 * memory of the handle, zeroed when a PID policy is installed = the agent's constructor,
 * baseline_agents.py:55-57).  It persists across launches and across nig_reset -- upstream never
 * resets the integral either (baseline_agents.py:61-80) -- so a rollout cut into several launches
 * equals one launch. */
int nig_set_policy(nig_handle *h, const nig_policy *policy /* host */, void *stream);

/*
 * n_steps closed-loop steps per lane in one launch: action = policy(observation), then
 * IndustrialEnv.step, state in registers.  Optional per-step outputs (any subset):
 *   obs_out   float row-major [n_steps][B][S]: the observation the policy acted on (D4RL
 *             'observations'), step k at obs_out + k*obs_step_stride
 *   act_out   float [n_steps][A][ld_act]: the policy's action (after its own clip, before the
 *             env's), step k at act_out + k*act_step_stride
 *   reward_out / flags_out   row of step k at base + k*out_stride (0 = overwrite)
 * Frozen lanes (finished, no auto-reset) write NIG_FLAG_INACTIVE and leave obs/act rows untouched.
 * This is the loop of utils.evaluate_with_safety (utils.py:80-112) and of the get_dataset
 * generators, with the agent on the device.
 */
int nig_rollout_policy(nig_handle *h, int32_t n_steps, float *reward_out, uint32_t *flags_out,
                       int64_t out_stride, float *obs_out, int64_t obs_step_stride,
                       float *act_out, int64_t ld_act, int64_t act_step_stride, void *stream);

/*
 * Closed-loop rollouts with the reference agents' deterministic ACTOR on the device: a
 * (S -> 256 -> 256 -> A) ReLU MLP with a tanh head (agents/networks.py:47-70,125-144;
 * cql.py:339-343) evaluated with float32-input MFMA (v_mfma_f32_32x32x2_f32: an exact k-ordered
 * float32 fma chain) and fused with IndustrialEnv.step in one kernel: 32 envs per wavefront, the
 * hidden activations never leave the accumulator registers (an accumulator tile is the next
 * layer's B operand as it stands).  Weights are host pointers, row-major [in][out] (the layout of
 * a Flax Dense kernel); the library re-orders them once into the MFMA operand stream and keeps
 * that copy in device memory it owns.  hidden must be 256.  The tanh head runs on K = 1 MFMAs sized
 * to the env's action count (v_mfma_f32_4x4x1 up to four actions, v_mfma_f32_16x16x1 up to sixteen);
 * envs with an odd state dim or more than sixteen actions are refused (NIG_ERR_UNSUPPORTED).
 */
int nig_set_mlp_policy(nig_handle *h, int32_t hidden, const float *W1, const float *b1, const float *W2,
                       const float *b2, const float *W3, const float *b3, void *stream);

/* As nig_rollout_policy, with the MLP actor installed by nig_set_mlp_policy. */
int nig_rollout_mlp(nig_handle *h, int32_t n_steps, float *reward_out, uint32_t *flags_out,
                    int64_t out_stride, float *obs_out, int64_t obs_step_stride,
                    float *act_out, int64_t ld_act, int64_t act_step_stride, void *stream);

/*
 * Host-buffer forms for SMALL batches -- the single-env drop-in classes (env.reset() / env.step()
 * return NumPy arrays, base.py:133-213).  Inputs are host arrays, results are copied back to host
 * arrays; the library stages through pinned memory it owns and synchronises `stream` once before
 * returning.  PCIe-/launch-latency bound by construction (tens of microseconds per call); use the
 * device-pointer entry points above for throughput.
 *   actions     float  [A][B] host            step_noise  double [k_step][B] host or NULL (fast mode)
 *   init_noise  double [k_reset][B] host or NULL
 *   state_out   float  [S][B] host (new state == observation)
 *   reward64_out double [B] host,  flags_out uint32 [B] host
 */
int nig_reset_host(nig_handle *h, const double *init_noise, float *state_out, void *stream);
int nig_step_host(nig_handle *h, const float *actions, const double *step_noise, float *state_out,
                  double *reward64_out, uint32_t *flags_out, void *stream);
/* the same with float64 actions double [A][B] (nig_step64's arithmetic) */
int nig_step_host64(nig_handle *h, const double *actions, const double *step_noise, float *state_out,
                    double *reward64_out, uint32_t *flags_out, void *stream);

/*
 * Mixed batch: several environment types in ONE padded structure-of-arrays batch (BASELINE config
 * "all 7 envs mixed-batch, heterogeneous state dims, padded SoA").  Lanes are grouped in contiguous
 * segments, one env type each, every segment starting on a multiple of 256 lanes (wavefronts and
 * blocks are homogeneous); all segments share ONE observation matrix float [S_max][ld] (rows >= S of
 * a segment stay zero), and actions / per-step outputs use the same column layout ([A_max][ld],
 * [n_steps][ld]).  The reference has no vectorised env at all: this stands in for N_total calls of
 * utils.make with different ids stepped by one loop (performance_benchmark.py:106-133 over a list of envs).
 * Global lane index of column c = env_index0 + c (keys the generator: results do not depend on the
 * other segments).  The fused rollout over the whole mixed batch is ONE kernel launch: a block finds
 * its segment in a table passed with the launch, the most expensive envs' blocks are issued first.
 */
#define NIG_MIXED_MAX_SEGMENTS 12
typedef struct nig_mixed nig_mixed;
typedef struct nig_mixed_info {
    int32_t n_segments, state_dim_max, action_dim_max, reserved;
    int64_t lanes;                                   /* sum of the segment counts                       */
    int64_t ld;                                      /* columns of the padded matrices (256-aligned)    */
    int32_t env[NIG_MIXED_MAX_SEGMENTS];             /* env id of segment k                             */
    int64_t offset[NIG_MIXED_MAX_SEGMENTS];          /* first column of segment k                       */
    int64_t count[NIG_MIXED_MAX_SEGMENTS];           /* lanes of segment k                              */
} nig_mixed_info;

/* flags as for nig_create (apply to every segment); seed / env_index0 as for nig_create */
int nig_create_mixed(int32_t n_segments, const int32_t *env_ids, const int64_t *counts, int device, uint64_t seed,
                     uint64_t env_index0, uint32_t flags, nig_mixed **out);
int nig_mixed_destroy(nig_mixed *m);
int nig_mixed_get_info(const nig_mixed *m, nig_mixed_info *out);
float *nig_mixed_state(const nig_mixed *m);                 /* device float [S_max][ld], library-owned          */
nig_handle *nig_mixed_segment(const nig_mixed *m, int32_t k);   /* borrowed handle of segment k: tallies, counters,
                                                               constraint mask, nig_step ... all per-handle calls */
int nig_mixed_reset(nig_mixed *m, void *stream);            /* IndustrialEnv.reset for every lane, fast mode    */
int nig_mixed_fill_actions(nig_mixed *m, uint32_t t, float *actions /* [A_max][ld] */, void *stream);
/* IndustrialEnv.step for every lane (fast mode): actions float [A_max][ld] (rows >= A of a segment ignored),
 * reward_out float [ld] / flags_out uint32 [ld] optional.  One step-kernel launch per segment on `stream`. */
int nig_mixed_step(nig_mixed *m, const float *actions, float *reward_out, uint32_t *flags_out, void *stream);
/* n_steps of IndustrialEnv.step for every lane of every segment in ONE launch; arguments as nig_rollout with
 * ld_act = ld: action_ring slot s at action_ring + s*slot_stride laid out [A_max][ld]; reward_out / flags_out
 * rows [ld] (both or neither), row of step k at base + k*out_stride (0 = overwrite). */
int nig_mixed_rollout(nig_mixed *m, int32_t n_steps, const float *action_ring, int64_t slot_stride, int32_t ring_len,
                      float *reward_out, uint32_t *flags_out, int64_t out_stride, void *stream);
/* The same single launch for handles the caller created itself (e.g. bound to its own matrix with
 * nig_bind_state): handle k uses columns [lane_offsets[k], lane_offsets[k] + batch_k) of the action ring and of
 * the output rows (pitch ld_act / out_stride as above).  All handles on one device, fast mode. */
int nig_rollout_mixed(nig_handle *const *handles, const int64_t *lane_offsets, int32_t n_handles, int32_t n_steps,
                      const float *action_ring, int64_t ld_act, int64_t slot_stride, int32_t ring_len,
                      float *reward_out, uint32_t *flags_out, int64_t out_stride, void *stream);
/* The same launch also returning the observation of every env.step (the obs of base.py:157-213's return tuple; what
 * get_dataset stores per transition, chemical_reactor.py:395-412, power_grid.py:234-249): obs_out float
 * [n_steps][S_max][ld_obs], the padded SoA layout of the state matrix per step -- handle k writes rows < S_k of its
 * columns [lane_offsets[k], +batch_k) of step s at obs_out + s*obs_step_stride (>= S_max*ld_obs, or 0 = overwrite),
 * rows >= S_k are not touched.  Needs reward_out and flags_out.  nig_mixed_rollout_obs: ld_obs = the batch's ld. */
int nig_rollout_mixed_obs(nig_handle *const *handles, const int64_t *lane_offsets, int32_t n_handles, int32_t n_steps,
                          const float *action_ring, int64_t ld_act, int64_t slot_stride, int32_t ring_len,
                          float *reward_out, uint32_t *flags_out, int64_t out_stride,
                          float *obs_out, int64_t ld_obs, int64_t obs_step_stride, void *stream);
int nig_mixed_rollout_obs(nig_mixed *m, int32_t n_steps, const float *action_ring, int64_t slot_stride, int32_t ring_len,
                          float *reward_out, uint32_t *flags_out, int64_t out_stride, float *obs_out, int64_t obs_step_stride,
                          void *stream);

/* Fill float [A][ld_act] with the synthetic uniform [-1,1) actions of stream
 * "action" for launch counter `t` (bench / parity workload generator). */
int nig_fill_actions(nig_handle *h, uint32_t t, float *actions, int64_t ld_act, void *stream);

/* Teacher forcing / checkpointing: copy state, counter words in or out (device pointers,
 * float [S][ld], uint32 [B]); either pointer may be NULL. */
int nig_set_state(nig_handle *h, const float *state, int64_t ld, const uint32_t *ctr, void *stream);
int nig_get_state(nig_handle *h, float *state, int64_t ld, uint32_t *ctr, void *stream);

/*
 * env.get_safety_metrics() for every lane (named by the reference README; defined as
 * the SafetyMetrics of the last step, base.py:94-124): int32 [5][ld_out] rows =
 * constraints_satisfied, total_constraints, violation_count, critical_violations,
 * and safety_score*total (== satisfied), decoded from a flag array of the last step.
 * total_constraints = the handle's ENABLED built-in constraints (nig_env_spec.n_constraints minus
 * those removed with nig_set_constraint_mask; 4 for AdvancedChemicalReactor); violation_count
 * includes NIG_FLAG_NVIOL_HI.
 */
int nig_get_safety_metrics(nig_handle *h, const uint32_t *flags, int32_t *out, int64_t ld_out,
                           void *stream);

/*
 * Reduce the per-lane tallies (NIG_F_TALLY) to one partial vector double[NIG_T_ROWS]
 * on the device (sum rows, min/max rows).  `partial_out` is a DEVICE pointer.  The
 * cross-GPU combine is an all-gather of these vectors (see parallel.py).
 */
int nig_reduce_tally(nig_handle *h, double *partial_out, void *stream);

/*
 * The path's one exchange (BASELINE north_star: "an RCCL all-reduce over xGMI only for the final return /
 * safety-violation reduction"), for hosts that do not go through torch.distributed: reduce the tallies of this
 * rank's handles (one, or the segments of a mixed batch) to one partial vector, ncclAllGather the partial vectors
 * of all ranks of `comm` (an ncclComm_t the caller created, one rank per GPU; NULL = single process, no collective)
 * and combine them in RANK ORDER on every rank: sums as fixed-order fp64 additions, exact integer counts, min / max
 * rows -- bit-identical on all ranks.  The 13 aggregates of evaluate_with_safety (utils.py:128-152) follow from
 * `out` (DEVICE double [NIG_T_ROWS]).  `scratch`: DEVICE doubles, at least (n_handles + 1 + ranks) * NIG_T_ROWS.
 * RCCL is taken from the copy already loaded in the process (dlsym), else librccl.so is dlopen'ed.
 */
int nig_reduce_metrics(nig_handle *const *handles, int32_t n_handles, void *nccl_comm, double *scratch,
                       int64_t scratch_doubles, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NIG_H */
