# Round 5, GPU call 15: where the HOST's time goes in one DDPM unlearning step (cProfile)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/hostprof_diffusion.py ddpm 2>&1 | grep -v amdgpu.ids | head -120 > gpurun_out/r05_hostprof_ddpm.txt
head -100 gpurun_out/r05_hostprof_ddpm.txt
