"""`python generate_mask.py --ckpt_path ... --classes 6 --device 0 [--nsfw True]` — command line of the reference's
SD/train-scripts/generate_mask.py:214-336 (same flags and defaults) in front of
`unlearn_saliency_amd.SD.train_scripts.generate_mask / generate_nsfw_mask`."""
import argparse

import _common


def build_parser():
    parser = argparse.ArgumentParser(prog="Train", description="train a stable diffusion model from scratch")
    parser.add_argument("--classes", help="class corresponding to concept to erase", type=str, required=False, default="6")
    parser.add_argument("--c_guidance", help="guidance of start image used to train", type=float, required=False, default=7.5)
    parser.add_argument("--batch_size", help="batch_size used to train", type=int, required=False, default=8)
    parser.add_argument("--epochs", help="epochs used to train", type=int, required=False, default=1)
    parser.add_argument("--lr", help="learning rate used to train", type=float, required=False, default=1e-5)
    parser.add_argument("--ckpt_path", help="ckpt path for stable diffusion v1-4", type=str, required=False,
                        default="models/ldm/stable-diffusion-v1/sd-v1-4-full-ema.ckpt")
    parser.add_argument("--config_path", help="config path for stable diffusion v1-4 inference", type=str, required=False,
                        default="configs/stable-diffusion/v1-inference.yaml")
    parser.add_argument("--diffusers_config_path", help="diffusers unet config json path", type=str, required=False,
                        default="diffusers_unet_config.json")
    parser.add_argument("--device", help="cuda devices to train on", type=str, required=False, default="4")
    parser.add_argument("--image_size", help="image size used to train", type=int, required=False, default=512)
    parser.add_argument("--num_timesteps", help="ddim steps of inference used to train", type=int, required=False, default=1000)
    # type=bool as in the reference: any non-empty string is True
    parser.add_argument("--nsfw", help="class or nsfw", type=bool, required=False, default=False)
    _common.add_batch_source_flags(parser)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    classes = int(args.classes)
    device = _common.device_of(args.device)
    from unlearn_saliency_amd.SD import train_scripts as TS
    model = TS.setup_model(args.config_path, args.ckpt_path, device, bf16=args.bf16,
                           resident_activations=args.resident_activations)
    data = _common.batches(args, device, {"forget": 3}, model)
    if args.nsfw:
        TS.generate_nsfw_mask(args.c_guidance, args.batch_size, args.epochs, args.lr, args.config_path, args.ckpt_path,
                              args.diffusers_config_path, device, args.image_size, args.num_timesteps, model=model,
                              forget_dl=data["forget"])
    else:
        TS.generate_mask(classes, args.c_guidance, args.batch_size, args.epochs, args.lr, args.config_path,
                         args.ckpt_path, args.diffusers_config_path, device, args.image_size, args.num_timesteps,
                         model=model, forget_dl=data["forget"])


if __name__ == "__main__":
    main()
