"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the IndustrialEnv.step() hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (neorl-industrial-gym_amd / libnig.so) never does.
"""
