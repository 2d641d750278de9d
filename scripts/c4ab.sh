for v in ""; do
  lib=$PWD/brdf_amd/libbrdf_hip$v.so
  BRDF_HIP_LIB=$lib timeout -k 10 200 python bench.py --workload c4 --entry bc_dif --steps 2 --warmup 1 --no-cpu > gpurun_out/r3u_c4$v.json 2>gpurun_out/r3u_c4$v.err
  python -c "
import json;d=json.load(open('gpurun_out/r3u_c4$v.json'));print('c4 bc_dif variant [$v]', d['ms_per_step'], d['result_sha256'])"
done
