#!/bin/bash
# tools/grun.sh <log name> -- runs tools/_gpu_cmd.sh on the GPU box with gpurun_out/r5 in place, output to gpurun_out/r5/<log name> and to the terminal
mkdir -p gpurun_out/r5
bash tools/_gpu_cmd.sh > "gpurun_out/r5/$1" 2>&1
cat "gpurun_out/r5/$1"
