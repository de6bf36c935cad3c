python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r06_gputests.log; echo "pytest rc=${PIPESTATUS[0]}" >> gpurun_out/r06_gputests.log
rm -rf gpurun_out/r06; tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1
tools/pmc_dual.sh > /dev/null 2>&1; cp gpurun_out/pmc_dual.txt gpurun_out/r06/pmc_dual.txt
tools/pmc_traffic.sh rows2 -- --config 3 --rows 2 > /dev/null 2>&1; cp gpurun_out/pmc_traffic_rows2.txt gpurun_out/r06/pmc_traffic_rows2.txt
tools/ab_deal.sh gpurun_out/r06/ab_deal.txt > /dev/null 2>&1
tail -3 gpurun_out/r06_gputests.log
