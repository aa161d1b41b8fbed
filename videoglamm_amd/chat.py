"""CLI demo with the reference's chat.py surface (R/chat.py:101-116 flags, :491-596 loop) on the MI355X path.

  python -m videoglamm_amd.chat --llava_version_or_path <hf dir> --video <frames dir | clip.npy> \
         --prompt_text "Please segment the dog" [--use_sam2_video_branch] [--precision bf16|fp32]

The container has no decord / cv2, so a "video" is a directory of frame images (sorted) or an .npy array
[T,H,W,3] uint8 already sampled at 1 fps (R/chat.py:66-98 does the sampling with decord).
"""
import argparse
import glob
import os

import numpy as np
import torch

from . import host, preproc
from .model import VideoGLaMMForCausalLM


def get_args():
    p = argparse.ArgumentParser()
    p.add_argument("--llava_version_or_path", type=str, required=True)
    p.add_argument("--vis_save_path", type=str, default="./vis_output/chat_output")
    p.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--model_max_length", type=int, default=2048)
    p.add_argument("--local_rank", type=int, default=0)
    p.add_argument("--use_mm_start_end", action="store_true")
    p.add_argument("--use_sam2_video_branch", action="store_true")
    p.add_argument("--base_model_type", type=str, default="vgpt|llama3_1", choices=["vgpt|phi3", "vgpt|llama3_1"])
    p.add_argument("--prompt_text", type=str, default="")
    p.add_argument("--video", type=str, required=True)
    p.add_argument("--max_new_tokens", type=int, default=512)
    return p.parse_args()


def load_frames(path, max_frames=64):
    """<= 64 frames like R/chat.py:386,392-395."""
    if path.endswith(".npy"):
        frames = list(np.load(path))
    else:
        from PIL import Image

        files = sorted(f for f in glob.glob(os.path.join(path, "*")) if f.lower().endswith((".jpg", ".jpeg", ".png")))
        frames = [np.array(Image.open(f).convert("RGB")) for f in files]
    if len(frames) > max_frames:
        idx = np.linspace(0, len(frames) - 1, max_frames, dtype=int)
        frames = [frames[i] for i in idx]
    return frames


def main():
    args = get_args()
    from transformers import AutoTokenizer

    tokenizer = AutoTokenizer.from_pretrained(args.llava_version_or_path, model_max_length=args.model_max_length, padding_side="right", use_fast=False)
    dtype = torch.bfloat16 if args.precision == "bf16" else torch.float32
    model = VideoGLaMMForCausalLM.from_pretrained(args.llava_version_or_path, torch_dtype=dtype, device=f"cuda:{args.local_rank}",
                                                  use_sam2_video_branch=args.use_sam2_video_branch)
    if "[SEG]" in tokenizer.get_vocab():
        model.config.seg_token_idx = model.cfg["seg_token_idx"] = tokenizer("[SEG]", add_special_tokens=False).input_ids[0]
    num_frames = int(os.environ.get("NUM_FRAMES", 16))
    base = args.base_model_type.split("|")[1]
    frames = load_frames(args.video)
    # row H1 on the device: the uint8 clip is uploaded once, the three inputs are made in HBM (preproc.py);
    # VG_HOST_PREPROCESS=1 keeps the PIL / numpy pipeline of host.py (same tensors, ~0.9 s per 8-frame clip on the host)
    if os.environ.get("VG_HOST_PREPROCESS", "0") == "1" or len({f.shape for f in frames}) != 1:
        images, context, sam, resize_list, original_size_list = host.preprocess_vision(frames, num_frames)
    else:
        preproc.DEVICE = f"cuda:{args.local_rank}"
        images, context, sam, resize_list, original_size_list = preproc.preprocess_vision(frames, num_frames)
    prompt = args.prompt_text or input("Please input your prompt: ")
    while True:
        input_ids = host.apply_for_chat(prompt, tokenizer, num_frames, base)
        output_ids, video_segments = model.inference(images, context, sam, input_ids, resize_list, original_size_list,
                                                     max_new_tokens=args.max_new_tokens,
                                                     use_sam2_video_branch=args.use_sam2_video_branch)
        print("text_output:", host.decode_text(output_ids, tokenizer))
        host.write_masks(video_segments[0], np.stack(frames), args.vis_save_path)
        if args.prompt_text:
            break
        prompt = input("Please input your prompt: ")


if __name__ == "__main__":
    main()
