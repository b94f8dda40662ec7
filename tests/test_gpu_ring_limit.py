"""-m gpu: the cooperating-wave kernels under the TEST-ONLY library variant whose ring waits are bounded
(csrc/nig_ring.hpp, -DNIG_RING_SPIN_LIMIT; VERDICT r03 #5, ADVICE r02/r03: "the ring protocols can only fail as a GPU hang").

1. The three-wave / paired-form tests run once under the variant: a protocol slip (or a toolchain that reorders a
   data / counter pair) would be an error code there -- and no wait may time out in a correct run.
2. The error path itself: with the variant's fault injection (producing waves stop posting after 7 steps) every family
   of ring kernels returns NIG_ERR_HIP naming the ring instead of hanging, and the handle works again afterwards.
Each part runs in a child interpreter (a process loads ONE libnig; the variant comes in through NIG_LIB_PATH)."""
import importlib.util
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def variant():
    spec = importlib.util.spec_from_file_location("_nig_build", os.path.join(ROOT, "neorl-industrial-gym_amd", "_build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b.build_variant("ringlimit")


def _child(variant, args, timeout=900):
    env = dict(os.environ, NIG_LIB_PATH=variant, NIG_NO_AUTOBUILD="1")
    return subprocess.run([sys.executable] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_ring_kernels_pass_their_tests_with_bounded_waits(variant):
    p = _child(variant, ["-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                         "tests/test_gpu_split.py", "tests/test_gpu_noise_rollout.py",
                         "tests/test_gpu_round3.py", "-k",
                         "split or three_wave or pair or paired or recorded_reference"])
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    assert " passed" in p.stdout and "ring protocol time-out" not in p.stdout


FAULT_SCRIPT = r'''
import sys, torch
import neorl_industrial_gym_amd as ni
L = ni._lib.lib()
assert L.nig_tune(ni._lib.TUNE_DIAG_RING_FAULT, 0) == 0, "not the bounded-wait variant"
ni.tune(split_blocks=256, wide_min_blocks=256)
cases = [("ChemicalReactor-v0", 1024, "open"), ("RobotAssembly-v0", 512, "open"), ("PowerGrid-v0", 512, "open"),
         ("ChemicalReactor-v0", 512, "policy")]
for name, B, kind in cases:
    env = ni.make_batched(name, B, autoreset=True, tally=True)
    env.reset()
    T = 40
    ring = torch.zeros(8, env.action_dim, env.ld, device="cuda")
    rew = torch.zeros(T, env.ld, device="cuda"); fl = torch.zeros(T, env.ld, dtype=torch.int32, device="cuda")
    if kind == "policy":
        env.set_policy(ni.behaviour_policy(name, "expert"))
        run = lambda: env.rollout_policy(T, rew, fl)
    else:
        run = lambda: env.rollout(T, ring, rew, fl)
    run(); torch.cuda.synchronize()                       # a correct launch: no error
    assert L.nig_tune(ni._lib.TUNE_DIAG_RING_FAULT, 1) == 0
    try:
        run()
        raise SystemExit(f"{name} {kind}: the injected fault was not reported")
    except ni._lib.NigError as e:
        msg = str(e)
        assert "ring protocol time-out" in msg and "polls" in msg, msg
        # the ring is named per kernel family (ADVICE r04): PowerGrid's paired form has {draws, slots released}, the three-wave
        # kernels {inputs, results, slots released}
        if name == "PowerGrid-v0":
            assert "'draws (producer -> stepper)'" in msg or "'slots released (stepper -> producer)'" in msg, msg
        else:
            assert "draws" not in msg and any(r in msg for r in ("'inputs (producer -> stepper)'", "'results (stepper -> recorder)'",
                                                                 "'slots released (consumer -> producer)'")), msg
        print("reported:", name, kind, "--", msg[:160])
    assert L.nig_tune(ni._lib.TUNE_DIAG_RING_FAULT, 0) == 0
    env.reset()
    run(); torch.cuda.synchronize()                       # the handle is usable again
    env.close()
print("ok")
'''


def test_a_ring_fault_is_an_error_code_not_a_hang(variant):
    p = _child(variant, ["-c", FAULT_SCRIPT], timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.stdout[-3000:], p.stderr[-3000:])
    assert p.stdout.count("reported:") == 4


def test_production_library_has_no_bounded_waits():
    import neorl_industrial_gym_amd as ni
    L = ni._lib.lib()
    assert L.nig_tune(ni._lib.TUNE_DIAG_RING_FAULT, 1) != 0 and b"test builds only" in L.nig_last_error()
