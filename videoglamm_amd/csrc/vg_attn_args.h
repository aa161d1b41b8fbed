// Arguments of the flash-attention kernels (vg_attention.hip, vg_attention_dma.hip).
#pragma once
#include "vg_common.h"

struct AttnArgs {
  const void* Q; const void* K; const void* V; void* O;
  int B, Hq, Hkv, Sq, Skv, D, causal;
  int64_t q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh;
  float scale;
  int nsplit, split_len;   // split-KV: blockIdx.x = q_tile * nsplit + split; partials go to `part`
  int xcd;                 // 1: (batch, head) blocks per XCD (see attn_kernel); needs (Hq * B) % 8 == 0
  float* part;             // [B, Hq, nsplit, Sq, D + 2]  (unnormalised O, running max m, running sum l)
  const int* skv_dev;      // optional: Skv = *skv_dev + Sq read on the device (graph-replayable decode step)
  int DV;                  // value / output head dim (== D except for the low-rank memory attention: vg_attention_dv)
  int fold;                // GQA fold: grid.y = Hkv and the G = Hq/Hkv query heads of a KV head become rows
                           // (row = g*Sq + q) of ONE query tile, so K/V are staged once per KV head (G*Sq <= tile)
};

// vg_attention_dma.hip: the LDS-DMA-staged kernel (bf16, head dim <= 128, no split / fold / window)
bool attn_dma_eligible(const AttnArgs& p);
int attn_dma_launch(const AttnArgs& p, hipStream_t st);
// ... and its head-dim-256 / 64-wide-value form with split-KV partials (vg_attention_dv)
bool attn_dma_dv_eligible(const AttnArgs& p);
int attn_dma_dv_launch(const AttnArgs& p, hipStream_t st);
