#!/usr/bin/env python3
"""BASELINE configs[0] alone: bench.measure_single_env (ChemicalReactor-v0, one env, 1 000 steps, reset on done) -- prints
steps/s.  (NIG_HOST_SPIN_US was the knob of a polling host wait tried in session 24: no difference, not kept.)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import neorl_industrial_gym_amd as ni
for _ in range(3):
    r = bench.measure_single_env(ni)
    print("NIG_HOST_SPIN_US=%s  %.0f steps/s  %.1f us per step" % (os.environ.get("NIG_HOST_SPIN_US", "(default)"), r["value"], r["us_per_step"]))
