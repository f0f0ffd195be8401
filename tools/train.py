#!/usr/bin/env python3
"""Entry point mirroring the reference's tools/train.py (/root/reference/tools/train.py:198-229).

The XE / self-critical training step (TransformerModel._forward, LanguageModelCriterion_UIC,
flat-bucket RCCL all-reduce) is the next row of the scope table and
is not built yet; this script builds the model exactly as the reference does and stops with a clear
error where the training forward would start, instead of silently training on another code path.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import captioning.models as models
    from boficap_amd.config import FULL
    model = models.setup(FULL.to_opt())
    print(f"built {type(model).__name__} with {sum(p.numel() for p in model.parameters())} parameters "
          f"({len(model.state_dict())} state_dict entries)")
    raise NotImplementedError("XE/RL training is not built in this round: see DESIGN.md §9 (next rows a14-a16)")


if __name__ == "__main__":
    main()
