"""HiFi-GAN decoder: latent z [B,192,T] + speaker embedding -> waveform [B,1,256*T].

Drop-in for /root/reference/module/model_component/decoder.py (``ResNetBlock`` :23-61,
``Decoder`` :64-131): same constructor arguments, ``forward`` signatures and ``state_dict``
keys (``conv1d_pre``, ``cond``, ``ups.{i}``, ``resblocks.{r}.convs{1,2}.{j}``,
``conv1d_post``).

Every convolution is one launch of the fp32-MFMA implicit-GEMM kernel with the
surrounding pointwise work fused in:
  * leaky-ReLU (decoder.py:49,54,115) is applied while the input tile is staged to LDS;
  * bias, the residual add (decoder.py:58), the sum over the three resblocks of a stage
    and the division by their number (decoder.py:119-125) are epilogues;
  * ConvTranspose1d (kernel 2*stride, padding stride/2) runs as a 2-tap polyphase
    convolution with stride*C_out GEMM rows, each output sample written once;
  * ``cond(g)`` (decoder.py:112) is a GEMV whose result becomes a per-utterance bias of
    ``conv1d_pre``;
  * weight norm is folded once per parameter version, not per forward.
The output stage (leaky-ReLU 0.01 -> conv 32->1 -> tanh, decoder.py:126-129) is a small
bandwidth-bound kernel of its own.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ... import native, packing
from ...engine import PackedConv, PlainConvParams, PlanCache, SpeakerTables, WeightNormConvParams, require_device

LRELU_SLOPE = 0.1


def get_padding(kernel_size, dilation=1):
    return int((kernel_size - 1) * dilation / 2)


class ResBlockPlan:
    def __init__(self, rb: "ResNetBlock", device):
        self.pairs = []
        # Resblocks run each (dilated conv, plain conv, residual) pair as one launch where that
        # is faster.  fp32: measured on the MI355X the fused kernel wins at 32 channels (118 vs
        # 96 TFLOP/s for the two-launch form) and loses at 64 (120 vs 126).  bf16 matrix inputs:
        # the convolutions are HBM-bound, so every width with a fused kernel (32/64/128) uses it.
        # VITS_AMD_PAIR_FUSION = "none" / "all" overrides the choice.
        from ...engine import precision
        mode = os.environ.get("VITS_AMD_PAIR_FUSION", "auto")
        prec = precision()
        pi = None if mode == "none" else native.pair_info(rb.channels, prec)
        k = rb.kernel_size
        # sliced fp32 (six matrix instructions per product), measured at 16 x 862 frames, one launch
        # against two, at 3 / 7 / 11 taps: 32 channels 0.40 / 0.60 / 0.88 ms against 0.64 / 0.83 / 1.05;
        # 64 channels 0.61 / 1.15 / 1.67 against 0.71 / 1.11 / 1.52 (the matrix work dominates there)
        wide_ok = (prec == native.PREC_BF16 or rb.channels <= 32 or mode == "all"
                   or (prec == native.PREC_F32_SPLIT and k <= 3))
        self.fused = (pi is not None and wide_ok and k % 2 == 1 and k <= 13
                      and all((k - 1) * d <= 52 for d in rb.dilation))
        for j, d in enumerate(rb.dilation):
            c1, c2 = rb.convs1[j], rb.convs2[j]
            if self.fused:
                def pk(c):
                    w = c.effective_weight().to(device)
                    if prec == native.PREC_BF16:
                        return packing.pack_conv_weight_bf16(w, pi.ck_bf16, pi.rows_per_block)
                    if prec == native.PREC_F32_SPLIT:
                        return packing.pack_conv_weight_split(w, pi.ck_split, pi.rows_per_block)
                    return packing.pack_conv_weight(w, pi.ck, pi.rows_per_block)

                def bs(c):
                    return c.bias.detach().to(device=device, dtype=torch.float32).contiguous()
                self.pairs.append((pk(c1), bs(c1), pk(c2), bs(c2), d))
            else:
                self.pairs.append((PackedConv(c1.effective_weight(), c1.bias, device, dilation=d),
                                   PackedConv(c2.effective_weight(), c2.bias, device, dilation=1)))
        self.taps = k
        # A resblock of short kernels as ONE launch (the three pairs chained in registers / LDS):
        # its pair launches wait on memory or pay a prologue and an epilogue per pair, and a
        # workgroup of the chained form recomputes only `reach` columns per side.  Longer kernels
        # (reach 36 / 60) lose more to the recompute than they save.  bf16 matrix inputs: where the
        # stage kernel takes one resblock per launch (128 / 256 channels; narrower stages run whole
        # through it).  fp32: 32 and 64 channels.
        self.chain = None
        mi = native.mrf_info(rb.channels, prec) if prec in (native.PREC_BF16, native.PREC_F32) else None
        reach = sum((k - 1) // 2 * (d + 1) for d in rb.dilation) if (k % 2 == 1 and k >= 3) else 1 << 30
        if (mi is not None and os.environ.get("VITS_AMD_MRF_FUSION", "1") != "0"
                and (self.fused or prec == native.PREC_F32)
                and reach <= self.chain_max_reach and len(rb.dilation) <= native.MRF_MAX_PAIRS
                and all((k - 1) // 2 * d <= 32 for d in rb.dilation)):
            same = (pi is not None and pi.rows_per_block == mi.rows_per_block and
                    (pi.ck_bf16 == mi.ck_bf16 if prec == native.PREC_BF16 else pi.ck == mi.ck))
            if self.fused and same:
                chain_pairs = self.pairs                # the pair kernel's packing is the chain's
            else:
                def pk(c):
                    w = c.effective_weight().to(device)
                    if prec == native.PREC_BF16:
                        return packing.pack_conv_weight_bf16(w, mi.ck_bf16, mi.rows_per_block)
                    return packing.pack_conv_weight(w, mi.ck, mi.rows_per_block)
                chain_pairs = [(pk(c1), c1.bias.detach().to(device=device, dtype=torch.float32).contiguous(),
                                pk(c2), c2.bias.detach().to(device=device, dtype=torch.float32).contiguous(), d)
                               for c1, c2, d in zip(rb.convs1, rb.convs2, rb.dilation)]
            self.chain = [(k, chain_pairs)]

    chain_max_reach = 16

    def _run_fused(self, x, tmp, run_buf, final_out, final_epi, acc_in, navg, before_final=None):
        # pairs ping-pong between the two scratch buffers: a pair's output may not land on its
        # input, whose halo columns neighbouring workgroups still read
        cur = x
        last = len(self.pairs) - 1
        for j, (w1, b1, w2, b2, d) in enumerate(self.pairs):
            if j < last:
                dst = run_buf if cur is not run_buf else tmp
                native.resblock_pair(cur, w1, b1, w2, b2, dst, taps=self.taps, dilation=d,
                                     epilogue=native.EPI_RESID, slope=LRELU_SLOPE)
                cur = dst
            else:
                if final_out.data_ptr() == cur.data_ptr():
                    raise native.NativeError("resblock: the output buffer aliases the last pair's input")
                if before_final is not None:
                    before_final()
                native.resblock_pair(cur, w1, b1, w2, b2, final_out, taps=self.taps, dilation=d,
                                     epilogue=final_epi, slope=LRELU_SLOPE, acc=acc_in, navg=navg)
        return final_out

    def run(self, x, tmp, run_buf, final_out, final_epi, acc_in=None, navg=1.0, before_final=None):
        """x -> resblock(x).  The last conv writes ``final_out`` with ``final_epi`` (plain
        residual, or residual + accumulation over resblocks); earlier pairs update ``run_buf``
        (and ``tmp``).  ``before_final`` is called just before the launch that touches
        ``final_out`` / ``acc_in`` (where a concurrent branch orders itself after its predecessor)."""
        if self.chain is not None:
            if final_out.data_ptr() == x.data_ptr():
                raise native.NativeError("resblock: the output buffer aliases the input")
            if before_final is not None:
                before_final()
            native.mrf_stage(x, final_out, self.chain, slope=LRELU_SLOPE,
                             acc=acc_in if final_epi != native.EPI_RESID else None,
                             navg=navg if final_epi == native.EPI_RESID_AVG else 1.0)
            return final_out
        if self.fused:
            return self._run_fused(x, tmp, run_buf, final_out, final_epi,
                                   acc_in if final_epi != native.EPI_RESID else None, navg, before_final)
        C = x.shape[1]
        cur = x
        last = len(self.pairs) - 1
        for j, (c1, c2) in enumerate(self.pairs):
            native.conv1d(cur, c1.w, tmp, c_in=C, c_out=C, taps=c1.taps, dilation=c1.dilation, pad=c1.pad,
                          tile=c1.tile, epilogue=native.EPI_STORE, bias=c1.bias,
                          act_in=True, in_slope=LRELU_SLOPE)
            if j < last:
                native.conv1d(tmp, c2.w, run_buf, c_in=C, c_out=C, taps=c2.taps, dilation=1, pad=c2.pad,
                              tile=c2.tile, epilogue=native.EPI_RESID, bias=c2.bias, res=cur,
                              act_in=True, in_slope=LRELU_SLOPE)
                cur = run_buf
            else:
                if before_final is not None:
                    before_final()
                native.conv1d(tmp, c2.w, final_out, c_in=C, c_out=C, taps=c2.taps, dilation=1, pad=c2.pad,
                              tile=c2.tile, epilogue=final_epi, bias=c2.bias, res=cur, out2=acc_in,
                              act_in=True, in_slope=LRELU_SLOPE, navg=navg)
        return final_out


class ResNetBlock(nn.Module):
    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.channels, self.kernel_size, self.dilation = channels, kernel_size, tuple(dilation)
        self.convs1 = nn.ModuleList(WeightNormConvParams(channels, channels, kernel_size) for _ in self.dilation)
        self.convs2 = nn.ModuleList(WeightNormConvParams(channels, channels, kernel_size) for _ in self.dilation)
        self._plans = PlanCache(self, lambda dev: ResBlockPlan(self, dev))

    def plan(self, device) -> ResBlockPlan:
        return self._plans.get(device)

    @torch.no_grad()
    def forward(self, x, x_mask=None):
        require_device(x, "ResNetBlock.forward")
        if x_mask is not None:
            raise native.NativeError("ResNetBlock: x_mask is never used on the inference path "
                                     "(decoder.py:122) and is not implemented in the HIP path")
        x = x.to(torch.float32).contiguous()
        return self.plan(x.device).run(x, torch.empty_like(x), torch.empty_like(x), torch.empty_like(x),
                                       native.EPI_RESID)


class DecoderPlan:
    def __init__(self, dec: "Decoder", device):
        pre = dec.conv1d_pre
        self.pre = PackedConv(pre.effective_weight(), None, device)
        self.cond_w = dec.cond.weight.detach().to(device)[:, :, 0].contiguous()
        self.cond_b = (dec.cond.bias.detach() + pre.bias.detach()).to(device).contiguous()
        self.speakers = SpeakerTables(self.cond_w, self.cond_b)     # cond(g) + conv1d_pre's bias, per utterance or per speaker
        self.ups = []
        for i, (s, k) in enumerate(zip(dec.deconv_strides, dec.deconv_kernel_sizes)):
            up = dec.ups[i]
            wt = up.effective_weight().to(device)
            poly = packing.transposed_to_polyphase(wt, s)
            order = packing.polyphase_row_order(wt.shape[1], s)
            self.ups.append((PackedConv(poly, up.bias, device, pad=1, row_order=None if order is None else order.to(device)),
                             s, wt.shape[1]))
        self.resblocks = [rb.plan(device) for rb in dec.resblocks]
        self.stages = [self._fused_stage(dec, i, device) for i in range(dec.num_deconvs)]
        self.post_w = dec.conv1d_post.weight.detach().to(device)[0].contiguous()    # [C, taps]


    @staticmethod
    def _fused_stage(dec, stage, device):
        """Weights of a whole multi-receptive-field stage packed for the one-launch kernel
        (bf16 matrix inputs, 32 / 64 channels, equal dilation lists, reach <= 64 samples), or
        None where the stage runs resblock by resblock.  VITS_AMD_MRF_FUSION=0 turns it off."""
        from ...engine import precision
        nrb = dec.num_resnet_blocks
        rbs = [dec.resblocks[stage * nrb + j] for j in range(nrb)]
        ch = rbs[0].channels
        if os.environ.get("VITS_AMD_MRF_FUSION", "1") == "0" or precision() != native.PREC_BF16:
            return None
        mi = native.mrf_info(ch, native.PREC_BF16)
        if mi is None or nrb > mi.max_blocks or any(len(rb.dilation) != len(rbs[0].dilation) for rb in rbs) \
                or len(rbs[0].dilation) > native.MRF_MAX_PAIRS:
            return None
        for rb in rbs:
            k = rb.kernel_size
            if k % 2 == 0 or k < 3 or sum((k - 1) // 2 * (d + 1) for d in rb.dilation) > 64 \
                    or any((k - 1) // 2 * d > 32 for d in rb.dilation):
                return None
        blocks = []
        for rb in rbs:
            pairs = []
            for j, d in enumerate(rb.dilation):
                c1, c2 = rb.convs1[j], rb.convs2[j]
                pairs.append((packing.pack_conv_weight_bf16(c1.effective_weight().to(device), mi.ck_bf16, mi.rows_per_block),
                              c1.bias.detach().to(device=device, dtype=torch.float32).contiguous(),
                              packing.pack_conv_weight_bf16(c2.effective_weight().to(device), mi.ck_bf16, mi.rows_per_block),
                              c2.bias.detach().to(device=device, dtype=torch.float32).contiguous(), d))
            blocks.append((rb.kernel_size, pairs))
        return blocks


class Decoder(nn.Module):
    def __init__(self, speaker_id_embedding_dim, in_z_channel=192, upsample_initial_channel=512,
                 deconv_strides=[8, 8, 2, 2], deconv_kernel_sizes=[16, 16, 4, 4],
                 resblock_kernel_sizes=[3, 7, 11],
                 resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]]):
        super().__init__()
        self.speaker_id_embedding_dim = speaker_id_embedding_dim
        self.in_z_channel = in_z_channel
        self.upsample_initial_channel = upsample_initial_channel
        self.deconv_strides = list(deconv_strides)
        self.deconv_kernel_sizes = list(deconv_kernel_sizes)
        self.resblock_kernel_sizes = list(resblock_kernel_sizes)
        self.resblock_dilation_sizes = [list(d) for d in resblock_dilation_sizes]
        self.num_deconvs = len(self.deconv_strides)
        self.num_resnet_blocks = len(self.resblock_kernel_sizes)
        for s, k in zip(self.deconv_strides, self.deconv_kernel_sizes):
            if k != 2 * s or s % 2:
                raise ValueError("the polyphase upsampling kernel needs kernel_size == 2*stride, stride even")
        c0 = upsample_initial_channel
        self.conv1d_pre = PlainConvParams(in_z_channel, c0, 7)
        self.cond = PlainConvParams(speaker_id_embedding_dim, c0, 1)
        self.ups = nn.ModuleList(
            WeightNormConvParams(c0 // 2 ** i, c0 // 2 ** (i + 1), k, transposed=True)
            for i, k in enumerate(self.deconv_kernel_sizes))
        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(self.num_deconvs):
            ch = c0 // 2 ** (i + 1)
            for k, d in zip(self.resblock_kernel_sizes, self.resblock_dilation_sizes):
                self.resblocks.append(ResNetBlock(ch, k, d))
        self.conv1d_post = PlainConvParams(ch, 1, 7, bias=False)
        self._plans = PlanCache(self, lambda dev: DecoderPlan(self, dev))

    # The resblocks of a stage are independent until their last convolution adds into the stage
    # output, so each can run on a stream of its own; the last launches are chained by events in
    # the reference's summation order (decoder.py:119-125), which keeps the result bit-identical.
    # Used with few utterances in flight, where one convolution does not fill 256 CUs.  At large
    # sizes the streams are worth 1-3 % (16 x 862 frames: fp32 66.7 -> 65.0 ms, bf16 inputs 12.26 ->
    # 12.05, sliced fp32 49.3 -> 48.8; VITS_AMD_BRANCH_STREAMS=on) but per-launch device times stop
    # being attributable to one kernel, so by default large batches run the resblocks one after another.
    concurrent_branch_frames = 2048         # use branch streams when batch * frames is at most this

    def _branches_concurrent(self, batch, frames):
        mode = os.environ.get("VITS_AMD_BRANCH_STREAMS", "auto")
        if mode in ("0", "off"):
            return False
        return mode in ("1", "on") or batch * frames <= self.concurrent_branch_frames

    def _run_branches_concurrently(self, plan, stage, xu, xs, epis):
        nrb = self.num_resnet_blocks
        main = torch.cuda.current_stream(xu.device)
        side_streams = native.branch_streams(xu.device, nrb - 1)
        fork = torch.cuda.Event()
        fork.record(main)
        prev_done = None
        for j in range(nrb):
            stream = main if j == 0 else side_streams[j - 1]
            if j > 0:
                stream.wait_event(fork)
            with torch.cuda.stream(stream):
                tmp, run_buf = torch.empty_like(xu), torch.empty_like(xu)
                if stream is not main:              # the caching allocator must not hand these out
                    tmp.record_stream(stream)       # again before this stream is done with them
                    run_buf.record_stream(stream)
                    xu.record_stream(stream)
                    xs.record_stream(stream)
                wait = (lambda e=prev_done, st=stream: st.wait_event(e)) if prev_done is not None else None
                plan.resblocks[stage * nrb + j].run(xu, tmp, run_buf, xs, epis[j], acc_in=xs, navg=float(nrb),
                                                    before_final=wait)
                prev_done = torch.cuda.Event()
                prev_done.record(stream)
        main.wait_event(prev_done)

    @torch.no_grad()
    def forward(self, z, speaker_id_embedded, z_mask=None):
        """``z_mask`` [B, 1, T] (not an argument of the reference): the decoder reads z * z_mask, the
        product the reference's callers form before the call (vits_generator.py:186,196), taken
        while the first convolution stages its input."""
        require_device(z, "Decoder.forward")
        dev = z.device
        plan = self._plans.get(dev)
        if z.dtype != torch.float32 or z.stride(2) != 1:
            z = z.to(torch.float32).contiguous()        # (a slice along time is read in place)
        B, _, T = z.shape
        c0 = self.upsample_initial_channel
        in_mask = None
        if z_mask is not None:
            in_mask = z_mask.to(torch.float32).expand(B, 1, z_mask.shape[-1])[:, :, :T].reshape(B, T).contiguous()

        pre_bias = plan.speakers.conditioning(speaker_id_embedded, B)
        x = torch.empty(B, c0, T, device=dev, dtype=torch.float32)
        native.conv1d(z, plan.pre.w, x, c_in=self.in_z_channel, c_out=c0, taps=plan.pre.taps, dilation=1,
                      pad=plan.pre.pad, tile=plan.pre.tile, epilogue=native.EPI_STORE,
                      bias=pre_bias.rows, bias_bstride=pre_bias.stride, bias_index=pre_bias.index, in_mask=in_mask)

        L = T
        nrb = self.num_resnet_blocks
        for i in range(self.num_deconvs):
            up, s, cout = plan.ups[i]
            cin = x.shape[1]
            L = L * s
            xu = torch.empty(B, cout, L, device=dev, dtype=torch.float32)
            native.conv1d(x, up.w, xu, c_in=cin, c_out=cout, taps=2, dilation=1, pad=1, tile=up.tile,
                          epilogue=native.EPI_UPSAMPLE, bias=up.bias, act_in=True, in_slope=LRELU_SLOPE,
                          up_stride=s)
            xs = torch.empty_like(xu)
            if plan.stages[i] is not None:
                # the whole stage in one launch: residual stream and the sum over kernel sizes in registers
                native.mrf_stage(xu, xs, plan.stages[i], slope=LRELU_SLOPE)
                x = xs
                continue
            epis = [native.EPI_RESID] + [native.EPI_RESID_ACC] * (nrb - 2) + [native.EPI_RESID_AVG] if nrb > 1 \
                else [native.EPI_RESID]
            if self._branches_concurrent(B, T) and nrb > 1:
                self._run_branches_concurrently(plan, i, xu, xs, epis)
            else:
                tmp = torch.empty_like(xu)
                run_buf = torch.empty_like(xu)
                for j in range(nrb):
                    plan.resblocks[i * nrb + j].run(xu, tmp, run_buf, xs, epis[j], acc_in=xs, navg=float(nrb))
            x = xs
        wav = torch.empty(B, 1, L, device=dev, dtype=torch.float32)
        native.post_conv_tanh(x, plan.post_w, wav, 0.01)
        return wav
