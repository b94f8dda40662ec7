# (round-1 script, kept as the record of that A/B: it swaps libnig.so in place; new A/Bs load variants through NIG_LIB_PATH, profiles/ab.sh)
cp neorl-industrial-gym_amd/libnig.so /tmp/libnig_orig.so; trap 'cp /tmp/libnig_orig.so neorl-industrial-gym_amd/libnig.so' EXIT
for r in 1 2; do for v in a b; do cp neorl-industrial-gym_amd/libnig_$v.so neorl-industrial-gym_amd/libnig.so; touch neorl-industrial-gym_amd/libnig.so
for cfg in "cr 65536" "ra 262144" "cr 1048576"; do read -r e b <<< "$cfg"; echo -n "$v $cfg step-api: "; timeout -k 10 100 python bench.py --env $e --batch $b --mode graph --steps 2000 --warmup 200 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('%.3e' % d['value'], d['roofline']['launch_us'])"; done; done; done
