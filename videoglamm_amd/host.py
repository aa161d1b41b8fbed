"""Host-side callers of the hot path (rows H1, H2, H4 of SURVEY.md §8a): frame sub-sampling and the three
preprocessed tensor streams, chat-prompt templating + <image> tokenisation, id post-processing and mask
writing.  CPU / numpy / PIL code, outside the timed region (the benchmark starts from device tensors);
the resizing backends the reference uses (cv2, torchvision, the CLIP processor from the HF hub) are not
installable here, so resizes go through PIL's bilinear/bicubic filters (see DESIGN.md "host rows").
"""
import os

import numpy as np
import torch

IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN, DEFAULT_VIDEO_TOKEN = "<image>", "<video>"

# ----------------------------------------------------------------------------------------------- H1
SAM_MEAN = torch.tensor([123.675, 116.28, 103.53]).view(-1, 1, 1)
SAM_STD = torch.tensor([58.395, 57.12, 57.375]).view(-1, 1, 1)
IV2_MEAN, IV2_STD = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])
CLIP_MEAN, CLIP_STD = np.array([0.48145466, 0.4578275, 0.40821073]), np.array([0.26862954, 0.26130258, 0.27577711])


def subsample_frames(frames, num_frames):
    """np.linspace sub-sampling to NUM_FRAMES — R/chat.py:422-427."""
    if len(frames) > num_frames:
        idx = np.linspace(0, len(frames) - 1, num_frames, dtype=int)
        return [frames[i] for i in idx]
    return list(frames)


def pad_or_truncate(frames, num_frames):
    """R/utils/enc_preprocessors.py:145-149: truncate, or repeat the last frame."""
    frames = list(frames[:num_frames])
    while len(frames) < num_frames:
        frames.append(frames[-1])
    return frames


def get_preprocess_shape(oldh, oldw, long_side):
    """ResizeLongestSide.get_preprocess_shape — R/model/segment_anything/utils/transforms.py."""
    scale = long_side * 1.0 / max(oldh, oldw)
    return int(oldh * scale + 0.5), int(oldw * scale + 0.5)


def _pil_resize(img_u8, hw, resample):
    from PIL import Image

    return np.array(Image.fromarray(img_u8).resize((hw[1], hw[0]), resample))


def sam_preprocess(frame_u8, img_size=1024):
    """sam_preprocess(model_type='sam2') — R/utils/sam_transforms.py:26-65: resize longest side to 1024
    (torchvision resize of a PIL image = PIL bilinear), normalise, bilinear stretch to 1024^2.
    -> (tensor [3,1024,1024] fp32, resize_shape)."""
    from PIL import Image

    th, tw = get_preprocess_shape(frame_u8.shape[0], frame_u8.shape[1], img_size)
    x = _pil_resize(frame_u8, (th, tw), Image.BILINEAR)
    x = torch.from_numpy(x).permute(2, 0, 1).contiguous()
    x = (x - SAM_MEAN) / SAM_STD
    x = torch.nn.functional.interpolate(x.unsqueeze(0), (img_size, img_size), mode="bilinear").squeeze(0)
    return x, (th, tw)


def cv_linear_taps(in_size, out_size, clamp_frac):
    """Taps of OpenCV's INTER_LINEAR resize for 8-bit images (imgproc/src/resize.cpp, resizeGeneric_ with INTER_RESIZE_COEF_BITS = 11):
    f = float32((d + 0.5) * scale - 0.5), s = floor(f), f -= s, taps = round_half_even({1 - f, f} * 2048).  Horizontal rule
    (clamp_frac): f = 0 at the borders (s < 0 or s >= in_size - 1); vertical rule: f kept, the two row indices clamped.
    -> (index0 int32 [out], index1 int32 [out], taps int32 [out, 2])."""
    scale = 1.0 / (float(out_size) / float(in_size))
    f = ((np.arange(out_size, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_frac:
        f = np.where((s < 0) | (s >= in_size - 1), np.float32(0.0), f).astype(np.float32)
    taps = np.stack([np.rint((np.float32(1.0) - f) * np.float32(2048.0)), np.rint(f * np.float32(2048.0))], axis=1).astype(np.int32)
    return np.clip(s, 0, in_size - 1).astype(np.int32), np.clip(s + 1, 0, in_size - 1).astype(np.int32), taps


def cv2_resize_linear_u8(img, hw):
    """cv2.resize(img, (w, h)) with the default INTER_LINEAR on an [H,W,C] uint8 image — what the reference's InternVideo2
    processor runs (R/model/videogpt_plus/model/internvideo/utils.py:124); a 2-tap, NON-antialiased filter in 11-bit fixed
    point, not Pillow's area-weighted triangle.  OpenCV is absent from this image: its published algorithm (4.x resize.cpp)
    is restated — same size: copy; exact 2x down-scale: the INTER_AREA fast path, (a + b + c + d + 2) >> 2; otherwise
    HResizeLinear (int32 row sums) + VResizeLinear (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2."""
    h, w = hw
    H, W = img.shape[:2]
    if (H, W) == (h, w):
        return img.copy()
    src = img.astype(np.int32)
    if H == 2 * h and W == 2 * w:
        return ((src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, xa = cv_linear_taps(W, w, True)
    y0, y1, yb = cv_linear_taps(H, h, False)
    rows = src[:, x0] * xa[None, :, 0, None] + src[:, x1] * xa[None, :, 1, None]          # [H, w, C] int32
    out = (((yb[:, 0, None, None] * (rows[y0] >> 4)) >> 16) + ((yb[:, 1, None, None] * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def iv2_preprocess(frame_u8, size=224):
    """VideoTrainProcessor.frames2tensor — R/.../internvideo/utils.py:105-143: cv2.resize (INTER_LINEAR) + (x/255 - mean)/std in
    numpy float64, cast to fp32 by the caller's .float()."""
    x = cv2_resize_linear_u8(frame_u8, (size, size)).astype(np.float64)
    x = (x / 255.0 - IV2_MEAN.reshape(1, 1, 3)) / IV2_STD.reshape(1, 1, 3)
    return torch.from_numpy(np.transpose(x, (2, 0, 1))).float()


def clip_preprocess(frame_u8, size=336):
    """CLIPImageProcessor(openai/clip-vit-large-patch14-336): bicubic resize of the short side to 336, centre crop,
    /255, CLIP mean/std — R/utils/enc_preprocessors.py:120-166."""
    from PIL import Image

    h, w = frame_u8.shape[:2]
    s = size / min(h, w)
    nh, nw = max(size, int(round(h * s))), max(size, int(round(w * s)))
    x = _pil_resize(frame_u8, (nh, nw), Image.BICUBIC)
    t, l = (nh - size) // 2, (nw - size) // 2
    x = x[t:t + size, l:l + size].astype(np.float64) / 255.0
    x = (x - CLIP_MEAN.reshape(1, 1, 3)) / CLIP_STD.reshape(1, 1, 3)
    return torch.from_numpy(np.transpose(x, (2, 0, 1))).float()


NUM_FRAMES = int(os.environ.get("NUM_FRAMES", 16))     # R/model/videogpt_plus/constants.py (the reference reads the same variable)


class SAM_v2_Preprocess:
    """R/utils/sam_transforms.py:77-79."""
    def preprocess(self, x):
        return sam_preprocess(np.asarray(x))


class EncPreprocessor_VideoGPTPlus:
    """R/utils/enc_preprocessors.py:106-166: a list of frames -> {'images': NUM_FRAMES InternVideo2 tensors, 'context_images': NUM_FRAMES CLIP
    tensors} (truncated / last frame repeated); a single image -> {'images': its CLIP tensor, 'context_images': None}."""
    def __init__(self, num_frames=None):
        self.num_frames = NUM_FRAMES if num_frames is None else num_frames
        self.frame_resolution_iv, self.frame_resolution_clip = 224, 336

    def preprocess(self, pil_images):
        if not isinstance(pil_images, list):
            return {"images": clip_preprocess(np.asarray(pil_images), self.frame_resolution_clip), "context_images": None}
        frames = [np.asarray(f) for f in pad_or_truncate(pil_images, self.num_frames)]
        return {"images": [iv2_preprocess(f, self.frame_resolution_iv) for f in frames],
                "context_images": [clip_preprocess(f, self.frame_resolution_clip) for f in frames]}


def precision_dtype(precision):
    """R/chat.py:437 `x.bfloat16() if precision == "bf16" else (x.half() if precision == "fp16" else x.float())`; "fp16" is bf16 on this
    build (chat.initialize_model_videogptplus warns about it once per load: there is no fp16 compute path)."""
    return torch.float32 if precision == "fp32" else torch.bfloat16


def _to_model(x, precision):
    x = x.to(precision_dtype(precision))
    return x.cuda(non_blocking=True) if torch.cuda.is_available() else x


def preprocess_vision(np_images, type="video", enc_preprocessor=None, sam_preprocessor=None, conv_generator=None, precision="fp16"):
    """preprocess_vision — R/chat.py:402-489, same parameters, same five return values in the same order.
    np_images: B x T x (H x W x C) uint8 arrays, batch of one (what load_video / load_image return).
    -> (enc_image, enc_context_image, image_sam, original_size_list, resize_list): lists of one [T',3,h,w] tensor (enc_context_image is None
    for type="image"), [(H, W)] of the source frames, [(h, w)] of the longest-side resize."""
    assert len(np_images) == 1, "Batch size must be 1"
    enc_preprocessor = enc_preprocessor or EncPreprocessor_VideoGPTPlus(getattr(conv_generator, "NUM_FRAMES", None))
    sam_preprocessor = sam_preprocessor or SAM_v2_Preprocess()
    if type == "video":
        frames = list(np_images[0])
        enc = enc_preprocessor.preprocess(subsample_frames(frames, getattr(conv_generator, "NUM_FRAMES", None) or enc_preprocessor.num_frames))
        enc_image = [_to_model(torch.stack(enc["images"], dim=0), precision)]
        ctx = enc["context_images"]
        enc_context_image = None if ctx is None else [_to_model(torch.stack(ctx, dim=0), precision)]
        original_size_list = [tuple(frames[0].shape[:2])]
        sam, shapes = zip(*[sam_preprocessor.preprocess(f) for f in frames])
        image_sam, resize_list = [_to_model(torch.stack(sam, dim=0), precision)], [shapes[0]]
    elif type == "image":
        assert len(np_images[0]) == 1, "Time dimension must be 1"
        image_np = np_images[0][0]
        enc = enc_preprocessor.preprocess(image_np)
        enc_image = [_to_model(enc["images"].unsqueeze(0), precision)]
        ctx = enc["context_images"]
        enc_context_image = None if ctx is None else [_to_model(ctx.unsqueeze(0), precision)]
        original_size_list = [tuple(image_np.shape[:2])]
        sam, shape = sam_preprocessor.preprocess(image_np)
        image_sam, resize_list = [_to_model(sam.unsqueeze(0), precision)], [shape]
    else:
        raise ValueError(f"type must be 'video' or 'image', got {type!r}")
    return enc_image, enc_context_image, image_sam, original_size_list, resize_list


# ----------------------------------------------------------------------------------------------- H2
TEMPLATES = {  # R/model/videogpt_plus/conversation.py:124-144
    "phi3": dict(system="<|system|>\nYou are a helpful AI assistant.", roles=("\n<|user|>\n", "\n<|assistant|>\n"), style="mpt", sep="<|end|>", sep2=None),
    "llama3_1": dict(system="A chat between a curious user and an artificial intelligence assistant. "
                            "The assistant gives helpful, detailed, and polite answers to the user's questions.",
                     roles=("USER", "ASSISTANT"), style="two", sep=" ", sep2="<|end_of_text|>"),
}


def get_prompt(template, messages):
    """Conversation.get_prompt for the MPT and TWO separator styles — conversation.py:27-75."""
    t = TEMPLATES[template]
    if t["style"] == "mpt":
        ret = t["system"] + t["sep"]
        for role, msg in messages:
            ret += role + msg + t["sep"] if msg else role
        return ret
    seps = [t["sep"], t["sep2"]]
    ret = t["system"] + seps[0]
    for i, (role, msg) in enumerate(messages):
        ret += role + ": " + msg + seps[i % 2] if msg else role + ":"
    return ret


def tokenizer_image_token(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX):
    """R/model/videogpt_plus/mm_utils.py:17-37: tokenise around '<image>' and splice -200 placeholders."""
    chunks = [tokenizer(c).input_ids for c in prompt.split(DEFAULT_IMAGE_TOKEN)]
    ids, offset = [], 0
    if len(chunks) > 0 and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id:
        offset = 1
        ids.append(chunks[0][0])
    sep = [image_token_index] * (offset + 1)
    inter = [e for pair in zip(chunks, [sep] * len(chunks)) for e in pair][:-1]
    for x in inter:
        ids.extend(x[offset:])
    return torch.tensor(ids, dtype=torch.long)


DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN = "<im_start>", "<im_end>"          # R/model/videogpt_plus/constants.py
DEFAULT_VID_START_TOKEN, DEFAULT_VID_END_TOKEN = "<vid_start>", "<vid_end>"


class ConvGenerator_VideoGPTPlus:
    """The inference half of R/utils/conv_generator.py:201-222 + ConvGenerator_Base.apply_for_chat (:86-131)."""
    NUM_FRAMES = NUM_FRAMES

    def __init__(self, use_mm_start_end=False, base_type="phi3", num_frames=None):
        if base_type not in TEMPLATES:
            raise ValueError("Invalid base_llm_type")
        self.use_mm_start_end, self.base_type = use_mm_start_end, base_type
        if num_frames is not None:
            self.NUM_FRAMES = num_frames

    def apply_for_chat(self, prompt_text, type="video", tokenizer=None):
        """-> input_ids [1, L] (on the GPU when there is one, like the reference's .cuda())."""
        if type == "video":
            prompt = DEFAULT_VIDEO_TOKEN + "\n" + prompt_text
            replace_token, vid_replace_token = DEFAULT_IMAGE_TOKEN, DEFAULT_IMAGE_TOKEN * self.NUM_FRAMES
            if self.use_mm_start_end:
                replace_token = DEFAULT_IM_START_TOKEN + replace_token + DEFAULT_IM_END_TOKEN
                vid_replace_token = DEFAULT_VID_START_TOKEN + vid_replace_token + DEFAULT_VID_END_TOKEN
            prompt = prompt.replace(DEFAULT_IMAGE_TOKEN, replace_token).replace(DEFAULT_VIDEO_TOKEN, vid_replace_token)
        elif type == "image":
            prompt = DEFAULT_IMAGE_TOKEN + "\n" + prompt_text
            if self.use_mm_start_end:
                prompt = prompt.replace(DEFAULT_IMAGE_TOKEN, DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN)
        else:
            raise ValueError(f"type must be 'video' or 'image', got {type!r}")
        roles = TEMPLATES[self.base_type]["roles"]
        ids = tokenizer_image_token(get_prompt(self.base_type, [(roles[0], prompt), (roles[1], "")]), tokenizer).unsqueeze(0)
        return ids.cuda() if torch.cuda.is_available() else ids


def apply_for_chat(prompt_text, tokenizer, num_frames=16, base_type="llama3_1", type="video"):
    """ConvGenerator_VideoGPTPlus(base_type).apply_for_chat as a function -> input_ids [1,L] on the host."""
    return ConvGenerator_VideoGPTPlus(False, base_type, num_frames).apply_for_chat(prompt_text, type, tokenizer).cpu()


# ----------------------------------------------------------------------------------------------- H4
def decode_text(output_ids, tokenizer):
    """R/chat.py:572-577: drop the -200 placeholders, decode."""
    ids = output_ids[0]
    ids = ids[ids != IMAGE_TOKEN_INDEX]
    return tokenizer.decode(ids, skip_special_tokens=False).replace("\n", "").replace("  ", " ")


def write_masks(video_segments, video_frames_np, save_dir):
    """write_masks — R/chat.py:26-64 (frames as JPG, masks as PNG, red overlay at 0.5 alpha), PIL instead of cv2."""
    from PIL import Image

    for t, pred in video_segments.items():
        os.makedirs(os.path.join(save_dir, "img_frames"), exist_ok=True)
        Image.fromarray(video_frames_np[t]).save(os.path.join(save_dir, "img_frames", f"frame_{t}.jpg"))
        for obj_id, m in pred.items():
            m = m > 0
            os.makedirs(os.path.join(save_dir, f"pred_masks_{obj_id}"), exist_ok=True)
            Image.fromarray((m * 255).astype(np.uint8)).save(os.path.join(save_dir, f"pred_masks_{obj_id}", f"mask_{t}.png"))
            os.makedirs(os.path.join(save_dir, "masked_images"), exist_ok=True)
            img = video_frames_np[t].copy()
            img[m] = (video_frames_np[t] * 0.5 + m[:, :, None].astype(np.uint8) * np.array([255, 0, 0]) * 0.5)[m]
            Image.fromarray(img).save(os.path.join(save_dir, "masked_images", f"masked_img_{t}_{obj_id}.jpg"))


def mask_iou(pred, ref):
    """IoU = sum(and) / sum(or) — R/eval_gcg_metrics.py:26-35 ('mask mIoU vs ref' of BASELINE.md)."""
    inter, union = np.logical_and(pred, ref).sum(), np.logical_or(pred, ref).sum()
    return float(inter) / float(union) if union > 0 else 1.0
