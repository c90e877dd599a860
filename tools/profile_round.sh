#!/bin/bash
# The round's closing measurements on ONE box (run from the repo root on the GPU box): tools/profile_round.sh <tag f32> <tag bf16>
# rocprofv3 passes of the 720p fp32 forward and of the 1080p B = 4 bf16 forward (tools/profile_passes.sh), merged summaries,
# then bench.py lines for the headline and the other configurations.  Everything lands under gpurun_out/.
A=${1:-r04_n}; B=${2:-r04_o}
tools/profile_passes.sh gpurun_out/$A > gpurun_out/${A}_passes.log 2>&1
python tools/profile_merge.py gpurun_out/$A gpurun_out/$A > gpurun_out/${A}_merge.log 2>&1
tools/profile_passes.sh gpurun_out/$B --height 1080 --width 1920 --batch 4 --dtype bf16 > gpurun_out/${B}_passes.log 2>&1
python tools/profile_merge.py gpurun_out/$B gpurun_out/$B > gpurun_out/${B}_merge.log 2>&1
rm -rf gpurun_out/$A/*/ gpurun_out/$B/*/ 2>/dev/null
python bench.py --steps 10 --warmup 2 > gpurun_out/${A}_bench.json 2> gpurun_out/${A}_bench.err
python bench.py --steps 10 --warmup 2 --dtype bf16 --no-cpu-baseline > gpurun_out/${A}_bench_bf16.json 2>> gpurun_out/${A}_bench.err
python bench.py --steps 6 --warmup 2 --height 1080 --width 1920 --batch 4 --dtype bf16 --no-cpu-baseline > gpurun_out/${B}_bench_1080p_bf16.json 2>> gpurun_out/${A}_bench.err
python bench.py --steps 6 --warmup 2 --height 1080 --width 1920 --batch 4 --no-cpu-baseline > gpurun_out/${B}_bench_1080p_f32.json 2>> gpurun_out/${A}_bench.err
head -3 gpurun_out/${A}_summary.txt; head -3 gpurun_out/${B}_summary.txt
