"""Training-step API of the reference, kept call-compatible:

  train_llm(args, epoch, in_audio, log_melspec, text_token_padded, target_dir_vec, vid_indices,
            model, discriminator, model_optim, dis_optimizer, accelerator) -> dict      (train_llm.py:9-98)
  train_iter_gan(args, epoch, in_text, in_audio, target_poses, vid_indices,
                 pose_decoder, discriminator, pose_dec_optim, dis_optim) -> dict         (train_gan.py:13-103)

Same order and count of generator / discriminator forwards (BatchNorm running statistics
and RNG draws are side effects the reference's training trajectory depends on), same loss
recipe (literal `epoch > 10` gate, `+1e-8` terms, beta = 0.05 / 0.1), same dict keys.
`mixed_precision("bf16")` (or `args.mixed_precision = "bf16"`) runs the forwards and losses of a step under
torch.autocast: library GEMMs in bf16 (BASELINE.json configs 2 and 4), HIP kernels, reductions and losses in fp32,
fp32 master weights and optimizer.  The default is the reference's fp32 (run_ted.py:275).
What differs (numerics-preserving): the two generator forwards whose outputs
the reference only ever uses detached run under `no_grad` (still in training mode), the
batch-independent prototype branch is computed once per step, and the scalar losses are
fetched with one asynchronous device->host copy that is started before the generator backward is enqueued
(`_LossFetch`) instead of up to five `.item()` syncs after it.
"""
import contextlib

import os

import torch
import torch.nn.functional as F

from . import ops as _ops


_MIXED = None


def mixed_precision(mode):
    """Select the step's compute precision for every later train_llm / train_iter_gan call: None / "no" = fp32
    (default), "bf16" = bf16 GEMMs under autocast.  Returns the previous setting."""
    global _MIXED
    if mode not in (None, "no", "fp32", "bf16"):
        raise ValueError(f"hopmi: unknown mixed_precision mode {mode!r}")
    prev, _MIXED = _MIXED, (None if mode in (None, "no", "fp32") else mode)
    return prev


def _amp(args, tensor):
    mode = getattr(args, "mixed_precision", None) or _MIXED
    if mode == "bf16" and tensor.is_cuda:
        return torch.autocast("cuda", dtype=torch.bfloat16)
    return contextlib.nullcontext()


# random draws go through these three so tests can replay the reference's CPU stream on the GPU
def _randn_like(t):
    return torch.randn_like(t)


def _randperm(n, device):
    return torch.randperm(n, device=device)


def add_noise(data):
    """train_llm.py:5-7."""
    return data + _randn_like(data) * 0.1


def _unwrap(m):
    return m.module if hasattr(m, "module") and not hasattr(m, "step_cache") else m


def _step_cache(model):
    m = _unwrap(model)
    return m.step_cache() if hasattr(m, "step_cache") else contextlib.nullcontext()


def _backward(accelerator, loss, module):
    """accelerator.backward(loss) (train_llm.py:34,85); a GradSync is also told which module the loss trains."""
    if getattr(accelerator, "takes_only", False):
        accelerator.backward(loss, only=(_unwrap(module),))
    else:
        accelerator.backward(loss)


def _regularisers(args, outputs, z_context, z_mu, z_logvar, out_rand, z_rand):
    """train_llm.py:59-73 / train_gan.py:68-81."""
    beta = 0.05
    pose_l1 = F.smooth_l1_loss(outputs / beta, out_rand.detach() / beta, reduction="none") * beta
    pose_l1 = pose_l1.sum(dim=1).sum(dim=1)
    pose_l1 = pose_l1.view(pose_l1.shape[0], -1).mean(1)
    z_l1 = F.l1_loss(z_context.detach(), z_rand.detach(), reduction="none")
    z_l1 = z_l1.view(z_l1.shape[0], -1).mean(1)
    div_reg = torch.clamp(-(pose_l1 / (z_l1 + 1.0e-5)), min=-1000).mean()
    kld = None
    if args.z_type == "speaker":
        kld = -0.5 * torch.mean(1 + z_logvar - z_mu.pow(2) - z_logvar.exp())
    return div_reg, kld


_CAPTURE = None     # a graph._Capture while a training-step graph is being recorded (see graph.GraphedTrainStep)


class _LossFetch:
    """The step's single device->host transfer, started as soon as every loss term exists (i.e. BEFORE the generator
    backward and the optimizer step are enqueued) into pinned memory, and waited for at the end of the step.  The
    values are the same floats; the host just no longer waits for the backward / optimizer kernels before it returns,
    so it issues the next step while the device finishes this one (the reference's `.item()` calls drain the device
    up to five times per step, train_llm.py:88-96).  The status words of persistent GRU launches ride along: those of
    this step's forwards and of the PREVIOUS step's backward, so a hand-off time-out surfaces at most one step late.

    While a step graph is being captured (`_CAPTURE`), the copy goes into the capture's pinned buffer and the graph is
    cut right behind it: a replay waits for that cut's event only, decodes with `decode()`, and returns while the
    backward / optimizer segment still runs."""

    def __init__(self, args, gan, huber, kld, div_reg, gen_error, dis_error):
        terms = [("loss", args.loss_regression_weight, huber)]
        if kld is not None:
            terms.append(("KLD", args.loss_kld_weight, kld))
        if div_reg is not None:
            terms.append(("DIV_REG", args.loss_reg_weight, div_reg))
        if gan:
            terms += [("gen", args.loss_gan_weight, gen_error), ("dis", 1.0, dis_error)]
        self.terms = [(k, w) for k, w, _ in terms]
        stacked = [t.detach().float().reshape(()) for _, _, t in terms]
        cap = _CAPTURE
        if cap is not None:
            self.status = cap.take_status()
        else:
            self.status = _ops.deferred_status() if stacked[0].is_cuda else None    # persistent-kernel hand-off status words
        if self.status is not None:
            stacked.append(self.status.to(stacked[0].device).float().reshape(()))
        dev = torch.stack(stacked)
        if cap is not None:
            self.host, self.event = cap.fetch_into(self, dev), None
        elif dev.is_cuda:
            self.host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
            self.host.copy_(dev, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.host, self.event = dev, None
        self.captured = cap is not None

    @staticmethod
    def decode(terms, vals, has_status):
        """train_llm.py:88-98.  `if kld:` / `if div_reg:` in the reference are truthiness tests on the tensors: a term
        that is exactly 0.0 is left out."""
        vals = list(vals)
        if has_status and vals.pop() != 0.0:
            # (the stack kernel's status word is sticky and its launch sequence number was not advanced by the launch that timed
            # out: its workspaces are zeroed IN PLACE -- a recorded step holds them by address, they are never released)
            _ops.stack_ws_reset()
            raise RuntimeError("hopmi: a persistent kernel (GRU recurrence / WaveNet stack) timed out waiting for a hand-off; "
                               "set HOPMI_GRU_PERSISTENT=0 to use per-step / per-layer launches")
        ret = {}
        for (k, wgt), v in zip(terms, vals):
            if k in ("KLD", "DIV_REG") and v == 0.0:
                continue
            ret[k] = wgt * v
        return ret

    def result(self):
        if self.captured:                      # nothing has run yet: the replay decodes (graph.GraphedTrainStep)
            return {}
        if self.event is not None:
            self.event.synchronize()
        return self.decode(self.terms, self.host.tolist(), self.status is not None)


# epoch <= 10: do not compute the discriminator score that train_llm.py:43-44 computes and never uses (same parameters, buffers
# and returned losses; tests/test_gpu_parity.py::test_train_llm_unused_score_elision)
ELIDE_UNUSED_SCORE = True
# the loss recipe of train_llm.py:46-79 as one fused op (ops.hop_losses) instead of ~75 tensor operations
FUSED_LOSSES = True
# generator step of the GAN phase: the discriminator's PARAMETER gradients are not computed.  train_llm.py:43,85 back-propagates
# gen_error through the discriminator into the generator; the gradients that pass leaves in the discriminator's own parameters
# on the way are never used -- only model_optim steps (:86), and the next discriminator step begins with dis_optimizer.zero_grad()
# (:17).  Same losses, same parameters and buffers of both networks after every step (bitwise: the data gradient does not depend
# on them; tests/test_gpu_parity.py::test_train_llm_unused_discriminator_grads_elision); what differs is the content of
# discriminator.parameters()[i].grad between the two optimizers' steps: the discriminator step's gradients instead of their sum
# with the generator step's.
ELIDE_UNUSED_D_GRADS = os.environ.get("HOPMI_ELIDE_D_GRADS", "1") != "0"
# discriminator step: the per-sample part of the discriminator (GRU, linears) once on the real and the generated batch side by side
PAIRED_DISCRIMINATOR = os.environ.get("HOPMI_PAIRED_D", "1") != "0"
# generator step: the graded forward and the regulariser's no-grad forward through Model.forward_pair (one decoder recurrence launch
# per layer for both); A/B: HOPMI_PAIRED_FWD=0
PAIRED_FORWARDS = os.environ.get("HOPMI_PAIRED_FWD", "1") != "0"


class _params_take_no_grad:
    """Within the block, `module`'s parameters do not require a gradient: a forward recorded here back-propagates to its inputs
    only (autograd decides at forward time what a backward will compute)."""

    def __init__(self, module, active=True):
        self.ps = [p for p in module.parameters() if p.requires_grad] if active else []

    def __enter__(self):
        for p in self.ps:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.ps:
            p.requires_grad_(True)
        return False


def train_llm(args, epoch, in_audio, log_melspec, text_token_padded, target_dir_vec, vid_indices,
              model, discriminator, model_optim, dis_optimizer, accelerator):
    pre_seq = target_dir_vec[:, 0:16]
    dis_error = None
    gan = epoch > 10 and args.loss_gan_weight > 0.0
    with _step_cache(model):
        if gan:                                                                # train_llm.py:15-36
            dis_optimizer.zero_grad()
            with _amp(args, target_dir_vec):
                with torch.no_grad():                                          # only used detached (:24)
                    outputs, *_ = model(in_audio, log_melspec, text_token_padded, pre_seq, vid_indices)
                real, fake = add_noise(target_dir_vec), add_noise(outputs.detach().float())
                # (a wrapped discriminator -- DistributedDataParallel, accelerate -- is called through its wrapper only: the wrapper's
                # forward is what arms its gradient synchronisation)
                plain = _unwrap(discriminator) is discriminator
                pair = getattr(discriminator, "forward_pair", None) if (PAIRED_DISCRIMINATOR and plain) else None
                if pair is not None:                                           # (nets.ConvDiscriminator: same scores, see there)
                    dis_real, dis_fake = pair(real, fake, text_token_padded)
                else:
                    dis_real = discriminator(real, text_token_padded)
                    dis_fake = discriminator(fake, text_token_padded)
                dis_error = torch.sum(-torch.mean(torch.log(dis_real.float() + 1e-8) + torch.log(1 - dis_fake.float() + 1e-8)))
            _backward(accelerator, dis_error, discriminator)
            dis_optimizer.step()

        model_optim.zero_grad()
        with _amp(args, target_dir_vec):
            # The graded forward and the diversity regulariser's no-grad forward (train_llm.py:42,58) as one call when the module
            # offers it (model.Model.forward_pair: the pose decoder's recurrences run once over both batches); the speaker permutation
            # is drawn where the reference draws it in the random streams (between the two forwards' noise draws).
            pair_fn = getattr(_unwrap(model), "forward_pair", None) if (
                PAIRED_FORWARDS and _ops.GRU_PAIR and args.z_type == "speaker" and args.loss_reg_weight > 0.0 and _unwrap(model) is model
                and in_audio.is_cuda and not model._forward_hooks and not model._forward_pre_hooks) else None      # (hooks see `forward` calls)
            out_rand = z_rand = None
            if pair_fn is not None:
                (outputs, z_context, z_mu, z_logvar), (out_rand, z_rand) = pair_fn(
                    in_audio, log_melspec, text_token_padded, pre_seq, vid_indices,
                    lambda: vid_indices[_randperm(vid_indices.shape[0], vid_indices.device)])
            else:
                outputs, z_context, z_mu, z_logvar = model(in_audio, log_melspec, text_token_padded, pre_seq, vid_indices)
            if epoch > 10 or not ELIDE_UNUSED_SCORE:
                # (not under a wrapper that counts on a gradient for every parameter of every forward it has seen, see above)
                with _params_take_no_grad(discriminator, epoch > 10 and ELIDE_UNUSED_D_GRADS and _unwrap(discriminator) is discriminator):
                    dis_output = discriminator(outputs, text_token_padded)
                gen_error = -torch.mean(torch.log(dis_output.float() + 1e-8))
            else:
                # train_llm.py:43-44 scores the output here in every epoch, and :81 leaves gen_error out of the loss and of
                # the returned dict until epoch 11: all that survives of the call is the BatchNorm statistics update
                discriminator.update_statistics(outputs)
                gen_error = None
            outputs = outputs.float()                                          # losses in fp32
            kld = div_reg = None
            if (args.z_type == "speaker" or args.z_type == "random") and args.loss_reg_weight > 0.0:
                if out_rand is None:
                    rand_vids = None
                    if args.z_type == "speaker":
                        rand_vids = vid_indices[_randperm(vid_indices.shape[0], vid_indices.device)]
                    with torch.no_grad():                                      # only used detached (:60,65)
                        out_rand, z_rand, _, _ = model(in_audio, log_melspec, text_token_padded, pre_seq, rand_vids)
                speaker = args.z_type == "speaker"
                if FUSED_LOSSES:
                    # train_llm.py:46-79: huber, diversity regulariser, KLD and their weighted sum, fused (ops.hop_losses)
                    loss, vals = _ops.hop_losses(outputs, target_dir_vec, out_rand.float(), z_context.detach().float(), z_rand.float(),
                                                 z_mu.float() if speaker else None, z_logvar.float() if speaker else None,
                                                 args.loss_regression_weight, args.loss_reg_weight, args.loss_kld_weight)
                    huber_loss, div_reg, kld = vals[0], vals[1], (vals[2] if speaker else None)
                else:
                    huber_loss = F.smooth_l1_loss(outputs / 0.1, target_dir_vec / 0.1) * 0.1
                    div_reg, kld = _regularisers(args, outputs, z_context.float(), z_mu.float(), z_logvar.float(),
                                                 out_rand.float(), z_rand.float())
                    loss = huber_loss * args.loss_regression_weight + div_reg * args.loss_reg_weight
                    if kld is not None:
                        loss = loss + kld * args.loss_kld_weight
            elif FUSED_LOSSES:
                loss, vals = _ops.hop_losses(outputs, target_dir_vec, w_reg=args.loss_regression_weight)
                huber_loss = vals[0]
            else:
                huber_loss = F.smooth_l1_loss(outputs / 0.1, target_dir_vec / 0.1) * 0.1
                loss = huber_loss * args.loss_regression_weight
            if epoch > 10:                                                     # literal gate, train_llm.py:81
                loss = loss + gen_error * args.loss_gan_weight
        fetch = _LossFetch(args, gan, huber_loss, kld, div_reg, gen_error, dis_error)
        _backward(accelerator, loss, model)
        model_optim.step()
    return fetch.result()


def train_iter_gan(args, epoch, in_text, in_audio, target_poses, vid_indices,
                   pose_decoder, discriminator, pose_dec_optim, dis_optim):
    warm_up_epochs = args.loss_warmup
    pre_seq = target_poses.new_zeros((target_poses.shape[0], target_poses.shape[1], target_poses.shape[2] + 1))
    pre_seq[:, 0:args.n_pre_poses, :-1] = target_poses[:, 0:args.n_pre_poses]
    pre_seq[:, 0:args.n_pre_poses, -1] = 1
    dis_error = None
    gan = epoch > warm_up_epochs and args.loss_gan_weight > 0.0
    if gan:                                                                    # train_gan.py:27-43 (no noise)
        dis_optim.zero_grad()
        with torch.no_grad():
            out_dir_vec, *_ = pose_decoder(pre_seq, in_text, in_audio, vid_indices)
        dis_real = discriminator(target_poses, in_text)
        dis_fake = discriminator(out_dir_vec.detach(), in_text)
        dis_error = torch.sum(-torch.mean(torch.log(dis_real + 1e-8) + torch.log(1 - dis_fake + 1e-8)))
        dis_error.backward()
        dis_optim.step()

    pose_dec_optim.zero_grad()
    out_dir_vec, z, z_mu, z_logvar = pose_decoder(pre_seq, in_text, in_audio, vid_indices)
    beta = 0.1
    huber_loss = F.smooth_l1_loss(out_dir_vec / beta, target_poses / beta) * beta
    dis_output = discriminator(out_dir_vec, in_text)
    gen_error = -torch.mean(torch.log(dis_output + 1e-8))
    kld = div_reg = None
    if (args.z_type == "speaker" or args.z_type == "random") and args.loss_reg_weight > 0.0:
        rand_vids = None
        if args.z_type == "speaker":
            rand_vids = vid_indices[_randperm(vid_indices.shape[0], vid_indices.device)]
        with torch.no_grad():
            out_rand, z_rand, _, _ = pose_decoder(pre_seq, in_text, in_audio, rand_vids)
        div_reg, kld = _regularisers(args, out_dir_vec, z, z_mu, z_logvar, out_rand, z_rand)
        loss = args.loss_regression_weight * huber_loss + args.loss_reg_weight * div_reg
        if kld is not None:
            loss = loss + args.loss_kld_weight * kld
    else:
        loss = args.loss_regression_weight * huber_loss
    if epoch > warm_up_epochs:
        loss = loss + args.loss_gan_weight * gen_error
    fetch = _LossFetch(args, gan, huber_loss, kld, div_reg, gen_error, dis_error)
    loss.backward()
    pose_dec_optim.step()
    return fetch.result()
