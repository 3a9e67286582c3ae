bash tools/run_profile.sh r05 > gpurun_out/r05_prof.log 2>&1; tail -45 gpurun_out/r05_prof.log
