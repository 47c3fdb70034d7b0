T=r03h; mkdir -p gpurun_out/$T; O=gpurun_out/$T
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err
SPH_NO_CG_FUSED_P=1 SPH_NO_DFSPH_FUSED_DIV=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_unfused.json 2> $O/bench_unfused.err
python - $O <<'PY'
import json,sys
for t in ('fused','unfused'):
    try:
        d=json.load(open(sys.argv[1]+'/bench_%s.json'%t)); e=d['extras']
        print(t, 'c2 %.4f' % d['ms_per_step'], 'c3 %.4f ms/step' % e['c3']['ms_per_step'], {k:round(v['avg_us'],1) for k,v in e['c3']['kernels'].items()})
        print('   c5 %.4f ms/step, %.2f cg it/step, %.2f us/it' % (e['c5']['ms_per_step'], e['c5']['cg_iterations_per_step'], e['c5']['us_per_cg_iteration']), {k:(v['launches_per_step'], round(v['avg_us'],1)) for k,v in e['c5']['kernels'].items() if k.startswith('cg')})
    except Exception as ex: print(t,'ERR',ex)
PY
cd _refdata 2>/dev/null && for s in final_scene4 final_scene0; do python ../tools/scene0_iterations.py --scene-file data/scenes/$s.json --steps 20 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print(d['scene'], 'ms/step %.2f' % d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if k.startswith('cg') or k.startswith('dfsph')})"; done
