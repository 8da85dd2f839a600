"""`python nsfw_removal.py --train_method full --mask_path mask/nude_0.5.pt --device 0` — command line of the
reference's SD/train-scripts/nsfw_removal.py:218-322 in front of `unlearn_saliency_amd.SD.train_scripts.nsfw_removal`
(BASELINE.json configs[4]: concept "nudity").

Reference quirks kept visible rather than reproduced: its `--lr` is declared `type=int` (so any `--lr 1e-5` on the
command line is rejected there; here it parses as float, default unchanged) and its `--device` default is "0,0", which
its own `int(args.device)` cannot convert (here: the first entry is used)."""
import argparse

import _common
from random_label import save_compvis


def build_parser():
    parser = argparse.ArgumentParser(prog="TrainESD",
                                     description="Finetuning stable diffusion model to erase concepts using ESD method")
    parser.add_argument("--train_method", help="method of training", type=str, required=True)
    parser.add_argument("--alpha", help="guidance of start image used to train", type=float, required=False, default=0.1)
    parser.add_argument("--batch_size", help="batch_size used to train", type=int, required=False, default=8)
    parser.add_argument("--epochs", help="epochs used to train", type=int, required=False, default=1)
    parser.add_argument("--lr", help="learning rate used to train", type=float, required=False, default=1e-5)
    parser.add_argument("--config_path", help="config path for stable diffusion v1-4 inference", type=str, required=False,
                        default="configs/stable-diffusion/v1-inference.yaml")
    parser.add_argument("--ckpt_path", help="ckpt path for stable diffusion v1-4", type=str, required=False,
                        default="models/ldm/stable-diffusion-v1/sd-v1-4-full-ema.ckpt")
    parser.add_argument("--mask_path", help="mask path for stable diffusion v1-4", type=str, required=False, default=None)
    parser.add_argument("--diffusers_config_path", help="diffusers unet config json path", type=str, required=False,
                        default="diffusers_unet_config.json")
    parser.add_argument("--device", help="cuda devices to train on", type=str, required=False, default="0,0")
    parser.add_argument("--image_size", help="image size used to train", type=int, required=False, default=512)
    parser.add_argument("--ddim_steps", help="ddim steps of inference used to train", type=int, required=False, default=50)
    _common.add_batch_source_flags(parser)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    device = _common.device_of(str(args.device).split(",")[0])
    from unlearn_saliency_amd.SD import train_scripts as TS
    model = TS.setup_model(args.config_path, args.ckpt_path, device, bf16=args.bf16,
                           resident_activations=args.resident_activations)
    data = _common.batches(args, device, {"forget": 3, "remain": 2}, model)
    model, losses = TS.nsfw_removal(args.train_method, args.alpha, args.batch_size, args.epochs, args.lr, args.config_path,
                                    args.ckpt_path, args.mask_path, args.diffusers_config_path, device, args.image_size,
                                    args.ddim_steps, model=model, forget_dl=data["forget"], remain_dl=data["remain"])
    name = (f"compvis-nsfw-mask-method_{args.train_method}-lr_{args.lr}" if args.mask_path
            else f"compvis-nsfw-method_{args.train_method}-lr_{args.lr}")  # nsfw_removal.py:78-82
    print("saved", save_compvis(model, name), "final loss", losses[-1] if losses else None)


if __name__ == "__main__":
    main()
