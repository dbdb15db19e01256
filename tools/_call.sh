bash tools/measure.sh r5g bench
