#!/usr/bin/env bash
# round-3 probe 1: what each class of ops costs INSIDE the pipelined step (FPD_WHATIF lowers the class to no-ops: timing only,
# results are wrong), base and variants interleaved on ONE box.  Output: gpurun_out/r03whatif/summary.txt
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/${WHATIF_OUT:-r03whatif}; mkdir -p $O
run() {  # name, FPD_WHATIF value, extra env
  FPD_WHATIF="$2" timeout 200 env $3 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/$1.json 2> $O/$1.err
  python -c "import json;d=json.load(open('$O/$1.json'));print('%-28s %7.3f ms/step' % ('$1', d['ms_per_step']))" 2>/dev/null || { echo "$1 FAILED"; tail -3 $O/$1.err; }
}
{
run base ""
run nowgrad nowgrad
run nowgrad_small nowgrad_small
run nowgrad_big nowgrad_big
run student_alone t_all
run student_alone_nowgrad t_all,nowgrad
run t_nobig t_big
run t_nomid t_mid
run t_nosmall t_small
run nobigconv nobigconv
run nobig nobig
run nomid nomid
run nosmall nosmall
run noapply noapply
run noew noew
run nobig_t_all nobig,t_all
run nosmall_t_all nosmall,t_all
run nomid_t_all nomid,t_all
run teacher_only nobig,nomid,nosmall,nowgrad
run base2 ""
run cap256 "" FPD_BNECK_BLOCKS=256
run cap256_student_nowgrad nowgrad FPD_BNECK_BLOCKS=256
timeout 300 python tools/probes/phase_times.py 2>/dev/null
} | tee $O/summary.txt
