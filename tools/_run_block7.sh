cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/hostprof_diffusion.py ddpm > gpurun_out/hostprof_ddpm.txt 2>&1
timeout 900 python tools/hostprof_diffusion.py sd > gpurun_out/hostprof_sd.txt 2>&1
head -3 gpurun_out/hostprof_ddpm.txt; head -3 gpurun_out/hostprof_sd.txt
