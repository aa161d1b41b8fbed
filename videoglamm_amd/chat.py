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
    """the reference's flags with the reference's defaults (R/chat.py:101-116) + the clip source and the token budget."""
    p = argparse.ArgumentParser()
    p.add_argument("--llava_version_or_path", type=str, required=True)
    p.add_argument("--vis_save_path", type=str, default="./vis_output/chat_output")
    p.add_argument("--precision", type=str, default="fp16", choices=["bf16", "fp16", "fp32"],
                   help="fp16 (the reference's default) runs the bf16 kernels: same 16-bit storage, wider exponent — there is no fp16 compute path")
    p.add_argument("--model_max_length", type=int, default=2048)
    p.add_argument("--vision_tower", type=str, default="openai/clip-vit-large-patch14", help="(chatunivi only in the reference; accepted, unused)")
    p.add_argument("--local_rank", type=int, default=0)
    p.add_argument("--load_in_8bit", action="store_true")
    p.add_argument("--load_in_4bit", action="store_true")
    p.add_argument("--use_mm_start_end", action="store_true")
    p.add_argument("--use_sam2_video_branch", action="store_true")
    p.add_argument("--base_model_type", type=str, default="vgpt|phi3", choices=["vgpt|phi3", "vgpt|llama3_1", "chatunivi"])
    p.add_argument("--prompt_text", type=str, default="")
    p.add_argument("--video", type=str, default="", help="frames directory or .npy clip; asked for interactively when empty (R/chat.py:513)")
    p.add_argument("--max_new_tokens", type=int, default=1024)     # R/chat.py:566
    return p.parse_args()


def initialize_model_videogptplus(model_base, precision="fp16", local_rank=0, load_in_8bit=False, load_in_4bit=False,
                                  use_sam2_video_branch=False, base_llm_type="phi3", tokenizer=None, **from_pretrained_kw):
    """R/chat.py:225-369, statement for statement on the façade: dtype choice, from_pretrained, tokenizer + "[SEG]", resize_token_embeddings,
    config ids, vision-module initialisation, tower movers, model dtype / device, eval.  -> (model, tokenizer).
    tokenizer: an already-built tokenizer (tests); default = AutoTokenizer.from_pretrained(model_base, use_fast=False)."""
    torch_dtype = torch.float32
    if precision == "bf16":
        torch_dtype = torch.bfloat16
    elif precision == "fp16":
        # the one flag where the drop-in changes numerics without being asked (fp16 is the reference's DEFAULT, R/chat.py:105,153): never
        # silent — a UserWarning on every load, and a refusal under VG_FP16_STRICT=1 for callers that must not run another number format
        import os
        import warnings
        msg = ("--precision fp16 (the reference's default) runs in bfloat16 on this build: the MI355X kernels compute in bf16 or fp32 "
               "(same 16-bit storage, 8-bit instead of 11-bit significand, fp32 exponent range); pass --precision bf16 to silence this, "
               "--precision fp32 for the parity mode")
        if os.environ.get("VG_FP16_STRICT", "0") == "1":
            raise NotImplementedError(msg + " [VG_FP16_STRICT=1]")
        warnings.warn(msg, UserWarning, stacklevel=2)
        torch_dtype = torch.bfloat16
    model_args = {"torch_dtype": torch_dtype}
    if load_in_4bit or load_in_8bit:
        model_args["load_in_4bit" if load_in_4bit else "load_in_8bit"] = True       # refused loudly by the model (bitsandbytes is CUDA-only)
    if base_llm_type not in ("phi3", "llama3_1"):
        raise ValueError("Invalid base_llm_type")
    model = VideoGLaMMForCausalLM.from_pretrained(model_base, low_cpu_mem_usage=False, use_sam2_video_branch=use_sam2_video_branch,
                                                  **model_args, **from_pretrained_kw)
    if tokenizer is None:
        from transformers import AutoTokenizer

        tokenizer = AutoTokenizer.from_pretrained(model_base, use_fast=False)
    tokenizer.pad_token = tokenizer.unk_token
    tokenizer.add_tokens("[SEG]")
    seg_token_idx = tokenizer.convert_tokens_to_ids("[SEG]")
    print("seg_token_idx: ", seg_token_idx)
    model.resize_token_embeddings(len(tokenizer))
    rows = model.P.sd["model.embed_tokens.weight"].shape[0]
    if not 0 <= seg_token_idx < rows:
        raise ValueError(f"[SEG] id {seg_token_idx} is outside the {rows}-row embedding table")
    model.config.seg_token_idx = seg_token_idx
    model.config.eos_token_id = tokenizer.eos_token_id
    model.config.bos_token_id = tokenizer.bos_token_id
    model.config.pad_token_id = tokenizer.pad_token_id
    if tokenizer.pad_token_id is None:      # llama3_1
        tokenizer.pad_token = tokenizer.eos_token
        tokenizer.pad_token_id = tokenizer.eos_token_id
    model.get_model().initialize_vision_modules(model.get_model().config)
    dev = local_rank if torch.cuda.is_available() else "cpu"
    model.get_model().get_vision_tower().to(dtype=torch_dtype, device=dev)
    model.get_model().get_image_vision_tower().to(dtype=torch_dtype, device=dev)
    if torch.cuda.is_available():
        model = (model.bfloat16() if torch_dtype == torch.bfloat16 else model.float()).cuda(local_rank)
    model.eval()
    return model, tokenizer


def load_video(path, max_frames=64):
    """load_video — R/chat.py:380-398: <= 64 frames, B x T x (H x W x C).  (No decord here: a frames directory or an .npy clip.)"""
    if path.endswith(".npy"):
        frames = list(np.load(path))
    else:
        from PIL import Image

        files = sorted(f for f in glob.glob(os.path.join(path, "*")) if f.lower().endswith((".jpg", ".jpeg", ".png")))
        frames = [np.array(Image.open(f).convert("RGB")) for f in files]
    if len(frames) > max_frames:
        idx = np.linspace(0, len(frames) - 1, max_frames, dtype=int)
        frames = [frames[i] for i in idx]
    return [frames]


def load_image(path):
    """load_image — R/chat.py:370-378: B x T x (H x W x C) with B = T = 1."""
    from PIL import Image

    return [[np.array(Image.open(path).convert("RGB"))]]


def main():
    args = get_args()
    if args.base_model_type.split("|")[0] != "vgpt":
        raise SystemExit("chatunivi compositions are outside this path (SURVEY §2): use --base_model_type 'vgpt|phi3' or 'vgpt|llama3_1'")
    base = args.base_model_type.split("|")[1]
    model, tokenizer = initialize_model_videogptplus(args.llava_version_or_path, args.precision, args.local_rank, args.load_in_8bit,
                                                     args.load_in_4bit, args.use_sam2_video_branch, base, device=f"cuda:{args.local_rank}")
    conv_generator = host.ConvGenerator_VideoGPTPlus(args.use_mm_start_end, base)
    enc_preprocessor, sam_preprocessor = host.EncPreprocessor_VideoGPTPlus(), host.SAM_v2_Preprocess()
    while True:
        path = args.video or input("Please input the image/video path: ")
        kind = "image" if os.path.splitext(path)[1].lower() in (".jpg", ".jpeg", ".png") else "video"        # R/chat.py:540
        np_images = load_image(path) if kind == "image" else load_video(path)
        # row H1 on the device: the uint8 clip is uploaded once, the three inputs are made in HBM (preproc.py);
        # VG_HOST_PREPROCESS=1 keeps the PIL / numpy pipeline of host.py (same tensors, ~0.9 s per 8-frame clip on the host)
        on_host = os.environ.get("VG_HOST_PREPROCESS", "0") == "1" or len({f.shape for f in np_images[0]}) != 1
        preproc.DEVICE = f"cuda:{args.local_rank}"
        enc_image, enc_context_image, image_sam, original_size_list, resize_list = (host if on_host else preproc).preprocess_vision(
            np_images, type=kind, enc_preprocessor=enc_preprocessor, sam_preprocessor=sam_preprocessor, conv_generator=conv_generator,
            precision=args.precision)
        prompt = args.prompt_text or input("Please input your prompt: ")
        input_ids = conv_generator.apply_for_chat(prompt, type=kind, tokenizer=tokenizer)
        output_ids, video_segments = model.inference(images=enc_image, context_images=enc_context_image, images_for_sam=image_sam,
                                                     input_ids=input_ids, resize_list=resize_list, original_size_list=original_size_list,
                                                     max_new_tokens=args.max_new_tokens,
                                                     use_sam2_video_branch=args.use_sam2_video_branch)
        text = host.decode_text(output_ids, tokenizer)
        print("text_output: ", text)
        save_dir = os.path.join(args.vis_save_path, os.path.basename(path.rstrip("/")).split(".")[0])
        os.makedirs(save_dir, exist_ok=True)
        host.write_masks(video_segments[0], np.stack(np_images[0]), save_dir)
        with open(os.path.join(save_dir, "caption.txt"), "w") as fh:       # R/chat.py:594-596
            fh.write(text)
        if args.prompt_text and args.video:
            break


if __name__ == "__main__":
    main()
