bash tools/measure.sh r5o bench
