bash tools/measure.sh r5h bench
