bash tools/measure.sh r5n tests stats pmc bench others matrix stage dist shards
