"""ORACLE (test infrastructure): end-to-end restatement of VideoGLaMM_SAM2.inference
(R/model/VideoGLaMM.py:560-879) on top of oracle/vlm.py and oracle/sam2.py, CPU fp32."""
import torch

from . import sam2, vlm


def inference(sd, cfg, images, context_images, images_for_sam, input_ids, original_size, max_new_tokens=32,
              use_sam2_video_branch=False, eos_token_id=None, return_logits=False, capture=None):
    """-> (output_ids [L+G], {frame: {obj: bool mask [H,W]}})  (batch of one, like the reference asserts).
    capture: optional dict that receives the [SEG] embeddings ("emb" [N,256]) and the logits before the threshold ("logits")."""
    ids, emb = vlm.generate(sd, cfg, images, context_images, input_ids, max_new_tokens, eos_token_id)
    if capture is not None:
        capture["emb"] = emb
    if emb.shape[0] == 0:
        return ids, ({} if not return_logits else None)
    p = "model.visual_model."
    if use_sam2_video_branch:
        logits, _ = sam2.video_branch(sd, p, cfg["sam2"], images_for_sam, emb, original_size)
        logits = torch.stack(logits)[:, :, 0]           # [T,N,H,W]
    else:
        logits, _ = sam2.framewise_branch(sd, p, cfg["sam2"], images_for_sam, emb, original_size)
        logits = torch.stack(logits)                    # [T,N,H,W]
    if capture is not None:
        capture["logits"] = logits
    if return_logits:
        return ids, logits
    seg = {t: {k: (logits[t, k] > 0).numpy() for k in range(logits.shape[1])} for t in range(logits.shape[0])}
    return ids, seg
