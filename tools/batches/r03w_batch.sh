# round 3, batch w: prologue / staging round-trip restructure -- correctness subset, then A/B against the previous build
O=gpurun_out/r03w; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_wcsph.py tests/test_hip_golden.py tests/test_big_golden.py tests/test_hip_solvers.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
A="--steps 100 --warmup 10 --no-cpu-baseline --no-extras"
for v in base main sb2pipe nopre; do
  if [ $v = main ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so; fi
  python bench.py $A > $O/c2_$v.json 2>/dev/null
  python bench.py $A --config c3 > $O/c3_$v.json 2>/dev/null
done
for v in base main; do
  if [ $v = main ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so; fi
  python tools/bench_c5.py --no-events > $O/c5_$v.txt 2>/dev/null
  python bench.py $A --method pcisph --motion-step 0 > $O/pci_$v.json 2>/dev/null
done
unset SPH_HIP_LIB
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    im = d.get('in_motion') or {}
    print(f.split('/')[-1].ljust(20), '%.4f rest' % d['ms_per_step'], ('%.4f motion' % im['ms_per_step']) if im else '', d['roofline']['kernel'], '%.1f us' % d['roofline']['avg_launch_us'],
          ' '.join('%s=%.1f' % (k, v['avg_us']) for k, v in (d['roofline'].get('all_kernels') or {}).items() if isinstance(v, dict) and v.get('avg_us', 0) > 20))
for f in sorted(glob.glob(sys.argv[1]+'/c5_*.txt')):
    print(f.split('/')[-1], open(f).read().strip().split('\n')[-1][:400])
PY
