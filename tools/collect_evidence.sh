# Side measurements quoted in DESIGN.md, written as text under gpurun_out/evidence/ (copy what is wanted into profiles/).
mkdir -p gpurun_out/evidence && cd $GRAFT_REPO_ROOT
E=gpurun_out/evidence
bash tools/ab_pipeline.sh > $E/ab_pipeline.txt 2>&1
bash tools/rehearse_scaling.sh > $E/scaling_rehearsal.txt 2>&1
timeout -k 10 300 python tools/gpu_stamps.py 2>&1 | grep -v amdgpu.ids > $E/workgroup_stamps.txt
timeout -k 10 600 python tools/gpu_clustered.py 2>&1 | grep -v amdgpu.ids > $E/clustered.txt
timeout -k 10 400 python tools/gpu_tail.py 2>&1 | grep -v amdgpu.ids > $E/tail_phases.txt
timeout -k 10 300 python tools/gpu_wide.py 2>&1 | grep -v amdgpu.ids > $E/wide_batches.txt
timeout -k 10 300 python tools/gpu_readbw.py 2>&1 | grep -v amdgpu.ids > $E/read_probe.txt
timeout -k 10 300 python tools/gpu_latency.py 2>&1 | grep -v amdgpu.ids > $E/host_api_latency.txt
timeout -k 10 120 python tools/gpu_sustained.py 2>&1 | grep -v amdgpu.ids > $E/sustained.txt
ls -la $E
