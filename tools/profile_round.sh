tools/profile_passes.sh gpurun_out/r03_d > gpurun_out/r03_d_passes.log 2>&1
python tools/profile_merge.py gpurun_out/r03_d gpurun_out/r03_d > gpurun_out/r03_d_merge.log 2>&1
tools/profile_passes.sh gpurun_out/r03_e --height 1080 --width 1920 --batch 4 --dtype bf16 > gpurun_out/r03_e_passes.log 2>&1
python tools/profile_merge.py gpurun_out/r03_e gpurun_out/r03_e > gpurun_out/r03_e_merge.log 2>&1
rm -rf gpurun_out/r03_d/*/ gpurun_out/r03_e/*/ 2>/dev/null
ls gpurun_out | head -30; head -3 gpurun_out/r03_d_summary.txt; head -3 gpurun_out/r03_e_summary.txt
