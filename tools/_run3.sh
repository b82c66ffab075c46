rm -f gpurun_out/nms_trace.txt
for b in 1 2 3; do bash tools/nms_trace.sh $PWD/gpurun_out/nms_trace.txt 600 720 1000 $b > /dev/null 2>&1; done
grep "====\|scan" gpurun_out/nms_trace.txt
