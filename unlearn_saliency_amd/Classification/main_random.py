"""`python main_random.py --unlearn RL --unlearn_epochs 10 --unlearn_lr 0.013 --num_indexes_to_replace 4500
--model_path ckpt --mask_path mask/with_0.5.pt --save_dir out` — same flags as the reference's
Classification/main_random.py; body in _driver.py."""
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import unlearn_saliency_amd.Classification  # noqa: F401
    __package__ = "unlearn_saliency_amd.Classification"

from ._driver import run


def main(argv=None):
    return run(argv, use_mask=True)


if __name__ == "__main__":
    main()
