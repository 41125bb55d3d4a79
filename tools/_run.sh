python -m pytest tests -m gpu -x -q 2>&1 | tail -12
for P in 0 1; do echo par_fk=$P; DCX_PAR_FK=$P python tools/mfma_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-60 | tee -a gpurun_out/r02_parfk_probe.txt; done
python tools/traj_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_parfk_traj_probe.txt
