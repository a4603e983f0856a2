#!/usr/bin/env python3
"""Entry point mirroring the reference CLI for the hot path: ``python main.py eval <config.yaml>`` (generation; reference main.py:54-66,
src/eval/workflow.py) and ``python main.py train <config.yaml>`` (multimodal SFT; src/train/mmsft/workflow.py).  export / download_data /
webui belong to subsystems outside the MI355X hot path."""
import sys


def main():
    if len(sys.argv) < 3 or sys.argv[1] not in ("eval", "train"):
        raise SystemExit("usage: python main.py eval|train <config.yaml>")
    if sys.argv[1] == "eval":
        from llamole_amd.eval import run_eval
        run_eval(sys.argv[2])
    else:
        from llamole_amd.train import run_train
        run_train(sys.argv[2])


if __name__ == "__main__":
    main()
