# A/B of two builds of the library (make -C brdf_amd/csrc variant SUFFIX=_base|_pair EXTRA=...): the default bench line, twice each
for v in base pair base pair; do
  BRDF_HIP_LIB=$PWD/brdf_amd/libbrdf_hip_$v.so timeout -k 10 200 python bench.py --steps 30 --warmup 5 > gpurun_out/ab_$v.json 2>/dev/null || exit 1
  python -c "import json;d=json.load(open('gpurun_out/ab_$v.json'));print('$v', d['ms_per_step'], d['roofline']['avg_launch_us'], d['bc_dif']['ms_per_step'], d['bc_dif']['avg_launch_us'], d['parity']['max_rel_err_params_vs_cpu_levmar'], d['bc_dif']['max_rel_err_params_vs_cpu_levmar'], d['config']['passes_per_fit'])"
  BRDF_HIP_LIB=$PWD/brdf_amd/libbrdf_hip_$v.so timeout -k 10 200 python bench.py --workload c4 --entry bc_dif --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('   c4 bc_dif', d['ms_per_step'], d['config']['failed_fits'])"
done
