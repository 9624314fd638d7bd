cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/fuzz2
python profiles/fuzz_campaign.py 5000 400000 > gpurun_out/fuzz2/default.log 2>&1; tail -n 1 gpurun_out/fuzz2/default.log
IPX_COMBINE=0 python profiles/fuzz_campaign.py 1000 410000 > gpurun_out/fuzz2/nocombine.log 2>&1; tail -n 1 gpurun_out/fuzz2/nocombine.log
IPX_SHORT_MAX_AVG=16 python profiles/fuzz_campaign.py 1000 420000 > gpurun_out/fuzz2/short16.log 2>&1; tail -n 1 gpurun_out/fuzz2/short16.log
IPX_LONG=1 IPX_SORT_TILE=4096 IPX_RANK=sort python profiles/fuzz_campaign.py 1000 430000 > gpurun_out/fuzz2/long1_tile4096_ranksort.log 2>&1; tail -n 1 gpurun_out/fuzz2/long1_tile4096_ranksort.log
IPX_FUZZ_BAM=1 python profiles/fuzz_campaign.py 400 440000 > gpurun_out/fuzz2/bam_records.log 2>&1; tail -n 1 gpurun_out/fuzz2/bam_records.log
