O=gpurun_out/r03o; mkdir -p $O
A="--steps 100 --warmup 10 --no-cpu-baseline --no-extras"
for v in main m28sb3 m28sb2; do
  if [ $v = main ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$PWD/sph_project_amd/variants/libsph_hip_$v.so; fi
  python bench.py $A --config c3 > $O/c3_$v.json 2>/dev/null
  python bench.py $A --method pcisph > $O/pci_$v.json 2>/dev/null
done
unset SPH_HIP_LIB
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    d=json.load(open(f)); print(f.split('/')[-1].ljust(18), '%.4f rest' % d['ms_per_step'], '%.4f motion' % d['in_motion']['ms_per_step'], 'fallback blocks', d['in_motion']['lds_fallback_blocks_last_step'], d['roofline']['kernel'], '%.1f us' % d['roofline']['avg_launch_us'])
PY
