#!/bin/bash
# Regenerates every classifier fixture of tests/golden/ for the current synthetic checkpoints (run after
# weights.make_synthetic or a kernel's summation order changes).  ~1.5 h on 8 AVX-512 cores; each step prints progress.
# The production 10k fixture resumes from its partial file.  ViT: `python tests/golden/make_classifier_fixtures.py vit`
# (its checkpoint generator is separate).
set -e
cd "$(dirname "$0")/../.."
make -C oracle
python tests/golden/make_fp32_module_fixture.py
python tests/golden/make_classifier_fixtures.py mfma_mc
python tests/golden/make_classifier_fixtures.py ens5
python tests/golden/make_classifier_fixtures.py mc
python tests/golden/make_classifier_fixtures.py 10k
python tests/golden/make_torchcpu_mc_fixture.py
python tests/golden/make_torchcpu_fixture.py
python tests/golden/make_classifier_fixtures.py mfma_10k
