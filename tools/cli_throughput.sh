#!/bin/bash
# End-to-end throughput of cuNVSMTrainModel on the Cranfield fixture (BASELINE configs[0] recipe: LSE, batch 4096,
# tanh): host batches over PCIe, loss read back every step, async prefetch. Prints batches/s and windows/s of the last
# epoch for the draw-for-draw host sampler and for the device sampler.   usage: tools/cli_throughput.sh [epochs]
cd "$(dirname "$0")/.."
EPOCHS=${1:-3}
for sampler in host device; do
  OUT=$(mktemp -d)
  ./cunvsm_amd/bin/cuNVSMTrainModel --word_repr_size 128 --entity_repr_size 256 --window_size 10 --num_random_entities 16 \
      --batch_size 4096 --nonlinearity tanh --bias_negative_samples --update_method full_adam --learning_rate 0.001 \
      --num_epochs $EPOCHS --seed 1 --sampler $sampler --v 1 --output $OUT/model tests/golden/cranfield/cranfield.trectext 2> $OUT/log
  echo "sampler=$sampler: $(grep -E 'Epoch #[0-9]+: duration' $OUT/log | tail -1 | sed 's/.*(\(.*batches\/second\)).*/\1/'); $(grep 'n-gram windows/second' $OUT/log | tail -1 | sed 's/.*: //')"
  rm -rf $OUT
done
