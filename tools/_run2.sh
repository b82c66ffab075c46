( timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "nms" 2>&1 | tail -8 ) > gpurun_out/t2.log 2>&1
rm -f gpurun_out/nms_trace.txt
bash tools/nms_trace.sh $PWD/gpurun_out/nms_trace.txt 600 720 1000 0 > /dev/null 2>&1
bash tools/nms_trace.sh $PWD/gpurun_out/nms_trace.txt 600 720 1000 1 > /dev/null 2>&1
bash tools/nms_trace.sh $PWD/gpurun_out/nms_trace.txt 320 480 50 0 > /dev/null 2>&1
bash tools/nms_trace.sh $PWD/gpurun_out/nms_trace.txt 320 480 50 1 > /dev/null 2>&1
bash tools/nms_trace.sh $PWD/gpurun_out/nms_trace.txt 720 1080 2000 1 > /dev/null 2>&1
cat gpurun_out/t2.log gpurun_out/nms_trace.txt
