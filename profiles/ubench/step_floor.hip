// step_floor.hip -- what does ONE step-API launch cost at the BASELINE headline size before any physics is computed?
// Build: hipcc --offload-arch=gfx950 -O3 -o step_floor step_floor.hip ; run on the GPU box.
// nig_plan_* replays n step_kernel launches from one hipGraph; at 65 536 ChemicalReactor lanes a replayed launch takes
// 5.0-5.2 us (bench.py step_api, profiles/r03/cr65536_driver_kernel_stats.csv) where the bytes it moves (124 B per
// lane) would take 1.0 us at the 8 TB/s peak.  This measures the floors under it with the same launch
// shape (256 blocks x 256 threads, one wave per SIMD, 250 kernel nodes per graph replay, each node depending on the
// one before through the stream order):
//   empty      a kernel that does nothing: the node-to-node boundary of the graph
//   roundtrip  the step kernel's memory shape without its arithmetic: load 16 dword rows (counter, 12 state rows, 3
//              action rows), ONE dependent use, store 15 rows (12 state, counter, reward, flags): one load round trip
//              + the store issue
//   chain2     the same with a second load batch that depends on the first (the step kernel's second round trip: the
//              generator's table lookups / the tally row of a finishing lane)
// each also as 250 plain stream launches (no graph).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int B = 65536, ROWS_IN = 16, ROWS_OUT = 15, NODES = 250;

__global__ void __launch_bounds__(256) k_empty(const float *, float *, const float *) {}

template <bool CHAIN2>
__global__ void __launch_bounds__(256) k_roundtrip(const float *in, float *out, const float *table)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    float v[ROWS_IN];
#pragma unroll
    for (int r = 0; r < ROWS_IN; ++r) v[r] = in[(size_t)r * B + i];
    float acc = 0.0f;
#pragma unroll
    for (int r = 0; r < ROWS_IN; ++r) acc += v[r];
    if constexpr (CHAIN2) {
        const unsigned idx = ((unsigned)__float_as_uint(acc) >> 9) & 767u;      // an address that depends on the first batch
        acc += table[idx * 4];
    }
#pragma unroll
    for (int r = 0; r < ROWS_OUT; ++r) out[(size_t)r * B + i] = v[r] + acc;
}

typedef void (*kern_t)(const float *, float *, const float *);

static int run(const char *name, kern_t k, float *a, float *b, float *tab, hipStream_t st)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // plain launches
    for (int w = 0; w < 2; ++w) for (int n = 0; n < NODES; ++n) hipLaunchKernelGGL(k, dim3(B / 256), dim3(256), 0, st, (n & 1) ? b : a, (n & 1) ? a : b, tab);
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int rep = 0; rep < 8; ++rep) for (int n = 0; n < NODES; ++n) hipLaunchKernelGGL(k, dim3(B / 256), dim3(256), 0, st, (n & 1) ? b : a, (n & 1) ? a : b, tab);
    CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
    float ms_plain; CHECK(hipEventElapsedTime(&ms_plain, e0, e1));
    // graph replay
    hipGraph_t g; hipGraphExec_t ge; hipStream_t cs;
    CHECK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
    for (int n = 0; n < NODES; ++n) hipLaunchKernelGGL(k, dim3(B / 256), dim3(256), 0, cs, (n & 1) ? b : a, (n & 1) ? a : b, tab);
    CHECK(hipStreamEndCapture(cs, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 4; ++w) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int rep = 0; rep < 16; ++rep) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
    float ms_graph; CHECK(hipEventElapsedTime(&ms_graph, e0, e1));
    printf("%-10s per launch: %6.2f us as graph nodes   %6.2f us as plain stream launches\n", name,
           ms_graph * 1e3 / (16 * NODES), ms_plain * 1e3 / (8 * NODES));
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g)); CHECK(hipStreamDestroy(cs));
    return 0;
}

int main()
{
    float *a, *b, *tab;
    CHECK(hipMalloc(&a, (size_t)ROWS_IN * B * 4)); CHECK(hipMalloc(&b, (size_t)ROWS_IN * B * 4)); CHECK(hipMalloc(&tab, 768 * 16));
    CHECK(hipMemset(a, 0, (size_t)ROWS_IN * B * 4)); CHECK(hipMemset(b, 0, (size_t)ROWS_IN * B * 4)); CHECK(hipMemset(tab, 0, 768 * 16));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    printf("65536 lanes, 256 blocks x 256 threads, %d dependent launches per replay; bytes per launch (roundtrip): %d B/lane\n",
           NODES, (ROWS_IN + ROWS_OUT) * 4);
    if (run("empty", k_empty, a, b, tab, st)) return 1;
    if (run("roundtrip", k_roundtrip<false>, a, b, tab, st)) return 1;
    if (run("chain2", k_roundtrip<true>, a, b, tab, st)) return 1;
    return 0;
}
