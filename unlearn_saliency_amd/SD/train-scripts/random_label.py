"""`python random_label.py --class_to_forget 0 --train_method full --mask_path mask/0/with_0.5.pt --device 0` —
command line of the reference's SD/train-scripts/random_label.py:210-328 in front of
`unlearn_saliency_amd.SD.train_scripts.certain_label`.  The unlearned U-Net is written as a CompVis-style state_dict
(`model.diffusion_model.*` keys) under models/<name>/<name>.pt like the reference's save_model (:159-207);
the Diffusers-layout U-Net (SD/convert.py) is written next to it."""
import argparse
import os

import _common


def build_parser():
    parser = argparse.ArgumentParser(prog="Train", description="train a stable diffusion model from scratch")
    parser.add_argument("--class_to_forget", help="class corresponding to concept to erase", type=str, required=True, default="0")
    parser.add_argument("--train_method", help="method of training", type=str, required=True)
    parser.add_argument("--alpha", help="guidance of start image used to train", type=float, required=False, default=0.1)
    parser.add_argument("--batch_size", help="batch_size used to train", type=int, required=False, default=8)
    parser.add_argument("--epochs", help="epochs used to train", type=int, required=False, default=5)
    parser.add_argument("--lr", help="learning rate used to train", type=float, required=False, default=1e-5)
    parser.add_argument("--ckpt_path", help="ckpt path for stable diffusion v1-4", type=str, required=False,
                        default="models/ldm/stable-diffusion-v1/sd-v1-4-full-ema.ckpt")
    parser.add_argument("--mask_path", help="mask path for stable diffusion v1-4", type=str, required=False, default=None)
    parser.add_argument("--config_path", help="config path for stable diffusion v1-4 inference", type=str, required=False,
                        default="configs/stable-diffusion/v1-inference.yaml")
    parser.add_argument("--diffusers_config_path", help="diffusers unet config json path", type=str, required=False,
                        default="diffusers_unet_config.json")
    parser.add_argument("--device", help="cuda devices to train on", type=str, required=False, default="4")
    parser.add_argument("--image_size", help="image size used to train", type=int, required=False, default=512)
    parser.add_argument("--ddim_steps", help="ddim steps of inference used to train", type=int, required=False, default=50)
    _common.add_batch_source_flags(parser)
    return parser


def save_compvis(model, name):
    """models/<name>/<name>.pt (CompVis state_dict) + the Diffusers-layout U-Net next to it, like the reference's
    save_model(save_compvis=True, save_diffusers=True) (random_label.py:159-207)."""
    import torch
    from unlearn_saliency_amd.SD.convert import savemodelDiffusers
    folder = f"models/{name}"
    os.makedirs(folder, exist_ok=True)
    torch.save(model.state_dict(), f"{folder}/{name}.pt")
    savemodelDiffusers(name, layers_per_block=len(model.model.diffusion_model.input_blocks) and
                       _layers_per_block(model.model.diffusion_model))
    return f"{folder}/{name}.pt"


def _layers_per_block(unet):
    """ResBlocks per resolution level = input blocks before the first Downsample (block 0 is conv_in)."""
    n = 0
    for blk in list(unet.input_blocks)[1:]:
        if any(type(m).__name__ == "Downsample" for m in blk):
            break
        n += 1
    return n


def main(argv=None):
    args = build_parser().parse_args(argv)
    classes = int(args.class_to_forget)
    device = _common.device_of(args.device)
    from unlearn_saliency_amd.SD import train_scripts as TS
    model = TS.setup_model(args.config_path, args.ckpt_path, device, bf16=args.bf16,
                           resident_activations=args.resident_activations)
    data = _common.batches(args, device, {"forget": 3, "remain": 2}, model)
    model, losses = TS.certain_label(classes, args.train_method, args.alpha, args.batch_size, args.epochs, args.lr,
                                     args.config_path, args.ckpt_path, args.mask_path, args.diffusers_config_path, device,
                                     args.image_size, args.ddim_steps, model=model, forget_dl=data["forget"],
                                     remain_dl=data["remain"])
    tag = "-mask" if args.mask_path else ""
    name = f"compvis-cl{tag}-class_{classes}-method_{args.train_method}-alpha_{args.alpha}-epoch_{args.epochs}-lr_{args.lr}"
    print("saved", save_compvis(model, name), "final loss", losses[-1] if losses else None)


if __name__ == "__main__":
    main()
