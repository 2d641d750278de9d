for v in base pair base pair; do
  BRDF_HIP_LIB=$PWD/brdf_amd/libbrdf_hip_$v.so timeout -k 10 200 python bench.py --steps 30 --warmup 5 > gpurun_out/ab_$v.json 2>/dev/null || exit 1
  python -c "import json;d=json.load(open('gpurun_out/ab_$v.json'));print('$v', d['ms_per_step'], d['roofline']['avg_launch_us'], d['bc_dif']['ms_per_step'], d['bc_dif']['avg_launch_us'], d['parity']['max_rel_err_params_vs_cpu_levmar'], d['config']['passes_per_fit'])"
done
