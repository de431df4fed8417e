#!/usr/bin/env python3
"""CGA fine-tune entry point (counterpart of the reference's cga.py): `freeze_for_n_epochs` epochs at min-lr in
which every weight farther than --boundaryRange from a rounding boundary gets a zero gradient and is restored after
optimizer.step().  Same flags as train.py plus --boundaryRange / --freeze_for_n_epochs, --qk_reparam_type 1."""
from ofq_amd.train_cli import main

if __name__ == "__main__":
    main(cga=True)
