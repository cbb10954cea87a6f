"""Trimodal baseline nets the HOP training step touches (reference:
`model/multimodal_context_net.py`).

`ConvDiscriminator` IS on the HOP training path (run_ted.py:276, train_llm.py:25-26,43).
`PoseGenerator` is kept for its call signature (`forward(pre_seq, in_text, in_audio,
vid_indices)`, multimodal_context_net.py:112) which `train_iter_gan` drives.  Both are
small recurrent/conv nets left on stock PyTorch-ROCm ops (MIOpen); SURVEY.md 8 rows a17 / b.
"""
import torch
import torch.nn as nn
from torch.nn.utils import weight_norm

from . import ops
from .model import WavEncoder, reparameterize


class Chomp1d(nn.Module):
    def __init__(self, chomp_size):
        super().__init__()
        self.chomp_size = chomp_size

    def forward(self, x):
        return x[:, :, :-self.chomp_size].contiguous()


class TemporalBlock(nn.Module):
    """EmbeddingSpaceEvaluator.py:44-76 (locuslab TCN block, weight-normed dilated convs)."""

    def __init__(self, n_inputs, n_outputs, kernel_size, stride, dilation, padding, dropout=0.2):
        super().__init__()
        self.conv1 = weight_norm(nn.Conv1d(n_inputs, n_outputs, kernel_size, stride=stride, padding=padding, dilation=dilation))
        self.chomp1, self.relu1, self.dropout1 = Chomp1d(padding), nn.ReLU(), nn.Dropout(dropout)
        self.conv2 = weight_norm(nn.Conv1d(n_outputs, n_outputs, kernel_size, stride=stride, padding=padding, dilation=dilation))
        self.chomp2, self.relu2, self.dropout2 = Chomp1d(padding), nn.ReLU(), nn.Dropout(dropout)
        self.net = nn.Sequential(self.conv1, self.chomp1, self.relu1, self.dropout1,
                                 self.conv2, self.chomp2, self.relu2, self.dropout2)
        self.downsample = nn.Conv1d(n_inputs, n_outputs, 1) if n_inputs != n_outputs else None
        self.relu = nn.ReLU()
        self.conv1.weight.data.normal_(0, 0.01)
        self.conv2.weight.data.normal_(0, 0.01)
        if self.downsample is not None:
            self.downsample.weight.data.normal_(0, 0.01)

    def forward(self, x):
        res = x if self.downsample is None else self.downsample(x)
        return self.relu(self.net(x) + res)


class TemporalConvNet(nn.Module):
    """EmbeddingSpaceEvaluator.py:78-94."""

    def __init__(self, num_inputs, num_channels, kernel_size=2, dropout=0.2):
        super().__init__()
        layers = []
        for i, out_c in enumerate(num_channels):
            d = 2 ** i
            in_c = num_inputs if i == 0 else num_channels[i - 1]
            layers.append(TemporalBlock(in_c, out_c, kernel_size, stride=1, dilation=d, padding=(kernel_size - 1) * d, dropout=dropout))
        self.network = nn.Sequential(*layers)

    def forward(self, x):
        return self.network(x)


class TextEncoderTCN(nn.Module):
    """multimodal_context_net.py:33-63."""

    def __init__(self, args, n_words, embed_size=300, pre_trained_embedding=None, kernel_size=2, dropout=0.3, emb_dropout=0.1):
        super().__init__()
        if pre_trained_embedding is not None:
            assert pre_trained_embedding.shape == (n_words, embed_size)
            self.embedding = nn.Embedding.from_pretrained(torch.FloatTensor(pre_trained_embedding), freeze=args.freeze_wordembed)
        else:
            self.embedding = nn.Embedding(n_words, embed_size)
        num_channels = [args.hidden_size] * args.n_layers
        self.tcn = TemporalConvNet(embed_size, num_channels, kernel_size, dropout=dropout)
        self.decoder = nn.Linear(num_channels[-1], 32)
        self.drop = nn.Dropout(emb_dropout)
        self.emb_dropout = emb_dropout
        self.decoder.bias.data.fill_(0)
        self.decoder.weight.data.normal_(0, 0.01)

    def forward(self, input):
        emb = self.drop(self.embedding(input))
        y = self.tcn(emb.transpose(1, 2)).transpose(1, 2)
        return self.decoder(y).contiguous(), 0


class PoseGenerator(nn.Module):
    """multimodal_context_net.py:66-172; note the argument order of forward differs from HOP.Model."""

    def __init__(self, args, pose_dim, n_words, word_embed_size, word_embeddings, z_obj=None):
        super().__init__()
        self.pre_length = args.n_pre_poses
        self.gen_length = args.n_poses - args.n_pre_poses
        self.z_obj = z_obj
        self.input_context = args.input_context
        if self.input_context == "both":
            self.in_size = 32 + 32 + pose_dim + 1
        elif self.input_context == "none":
            self.in_size = pose_dim + 1
        else:
            self.in_size = 32 + pose_dim + 1
        self.audio_encoder = WavEncoder()
        self.text_encoder = TextEncoderTCN(args, n_words, word_embed_size, pre_trained_embedding=word_embeddings,
                                           dropout=args.dropout_prob)
        self.speaker_embedding = None
        if self.z_obj:
            self.z_size = 16
            self.in_size += self.z_size
            if hasattr(self.z_obj, "n_words"):      # a Vocab (speaker ids); anything else = random noise z
                self.speaker_embedding = nn.Sequential(nn.Embedding(z_obj.n_words, self.z_size), nn.Linear(self.z_size, self.z_size))
                self.speaker_mu = nn.Linear(self.z_size, self.z_size)
                self.speaker_logvar = nn.Linear(self.z_size, self.z_size)
        self.hidden_size = args.hidden_size
        self.gru = nn.GRU(self.in_size, hidden_size=self.hidden_size, num_layers=args.n_layers, batch_first=True,
                          bidirectional=True, dropout=args.dropout_prob)
        self.out = nn.Sequential(nn.Linear(self.hidden_size, self.hidden_size // 2), nn.LeakyReLU(True),
                                 nn.Linear(self.hidden_size // 2, pose_dim))
        self._randn_like = torch.randn_like

    def forward(self, pre_seq, in_text, in_audio, vid_indices=None):
        text_feat_seq = audio_feat_seq = None
        if self.input_context != "none":
            audio_feat_seq = self.audio_encoder(in_audio)
            text_feat_seq, _ = self.text_encoder(in_text)
            assert audio_feat_seq.shape[1] == text_feat_seq.shape[1]
        z_mu = z_logvar = z_context = None
        if self.z_obj:
            if self.speaker_embedding:
                assert vid_indices is not None
                z_context = self.speaker_embedding(vid_indices)
                z_mu, z_logvar = self.speaker_mu(z_context), self.speaker_logvar(z_context)
                z_context = reparameterize(z_mu, z_logvar, self._randn_like)
            else:
                z_context = torch.randn(in_text.shape[0], self.z_size, device=in_text.device)
        if self.input_context == "both":
            in_data = torch.cat((pre_seq, audio_feat_seq, text_feat_seq), dim=2)
        elif self.input_context == "audio":
            in_data = torch.cat((pre_seq, audio_feat_seq), dim=2)
        elif self.input_context == "text":
            in_data = torch.cat((pre_seq, text_feat_seq), dim=2)
        elif self.input_context == "none":
            in_data = pre_seq
        else:
            raise AssertionError(self.input_context)
        if z_context is not None:
            in_data = torch.cat((in_data, z_context.unsqueeze(1).expand(-1, in_data.shape[1], -1)), dim=2)
        output, _ = self.gru(in_data, None)
        output = output[:, :, :self.hidden_size] + output[:, :, self.hidden_size:]
        output = self.out(output.reshape(-1, output.shape[2]))
        return output.reshape(in_data.shape[0], in_data.shape[1], -1), z_context, z_mu, z_logvar


class ConvDiscriminator(nn.Module):
    """multimodal_context_net.py:219-268: (B,34,P) -> sigmoid score (B,1)."""

    def __init__(self, input_size):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = 64
        self.pre_conv = nn.Sequential(
            nn.Conv1d(input_size, 16, 3), nn.BatchNorm1d(16), nn.LeakyReLU(True),     # LeakyReLU(True) == identity
            nn.Conv1d(16, 8, 3), nn.BatchNorm1d(8), nn.LeakyReLU(True),
            nn.Conv1d(8, 8, 3))
        self.gru = nn.GRU(8, hidden_size=self.hidden_size, num_layers=4, bidirectional=True, dropout=0.3, batch_first=True)
        self.out = ops.Linear(self.hidden_size, 1)
        self.out2 = ops.Linear(28, 1)

    @staticmethod
    def _conv3_cl(x, conv):
        """Valid Conv1d(k=3) on channels-last x (B,T,C): the three taps side by side, one GEMM
        (MIOpen's im2col path runs these tiny convs per sample: ~440 launches per training step)."""
        T = x.shape[1] - 2
        w = conv.weight.permute(0, 2, 1).reshape(conv.out_channels, -1)          # (O, 3*C), tap-major
        return ops.linear(torch.cat([x[:, 0:T], x[:, 1:T + 1], x[:, 2:T + 2]], dim=2), w, conv.bias)

    @staticmethod
    def _bn_cl(x, bn, training):
        """BatchNorm1d on channels-last x (B,T,C), torch semantics (batch stats over B and T): one launch forward, one backward
        (ops.batch_norm_cl; as tensor operations the layer was ~12 + ~15 launches at the launch floor, on 256 KB of data)."""
        return ops.batch_norm_cl(x, bn, training)

    def _pre_conv_cl(self, poses):
        """pre_conv (multimodal_context_net.py:226-234) without leaving the (B,T,C) layout; LeakyReLU(True)
        has slope 1.0, i.e. it is the identity."""
        x = self._bn_cl(self._conv3_cl(poses, self.pre_conv[0]), self.pre_conv[1], self.training)
        x = self._bn_cl(self._conv3_cl(x, self.pre_conv[3]), self.pre_conv[4], self.training)
        return self._conv3_cl(x, self.pre_conv[6])

    @torch.no_grad()
    def update_statistics(self, poses):
        """The only lasting effect of a training-mode `forward` whose score is discarded: the running-statistics update
        of the two BatchNorm1d layers of pre_conv (steps.train_llm, epoch <= 10: train_llm.py:43-44 computes the score and
        gen_error and :81 leaves them out of the loss)."""
        if not self.training:
            return
        x = self._bn_cl(self._conv3_cl(poses.detach(), self.pre_conv[0]), self.pre_conv[1], True)
        ops.batch_norm_cl_statistics(self._conv3_cl(x, self.pre_conv[3]), self.pre_conv[4])

    def _score(self, feat):
        """Everything behind pre_conv (multimodal_context_net.py:255-268): per sample, nothing couples the batch."""
        output = ops.gru_bidirectional(feat, self.gru, self.gru.dropout, self.training)
        output = output[:, :, :self.hidden_size] + output[:, :, self.hidden_size:]
        output = self.out(output.contiguous().view(-1, output.shape[2])).view(feat.shape[0], -1)
        return torch.sigmoid(self.out2(output))

    def forward(self, poses, in_text=None):
        # GEMM-shaped convs + the hand-written GRU recurrence (same cell as the decoder); like the generator this only
        # runs on a ROCm device (ops.gru_bidirectional raises for host tensors: no second, stock-torch path)
        return self._score(self._pre_conv_cl(poses))

    def forward_pair(self, poses_a, poses_b, in_text=None):
        """(D(poses_a), D(poses_b)) as the discriminator step scores the real and the generated batch (train_llm.py:25-26,
        train_gan.py:40-41): pre_conv -- whose BatchNorm layers take their statistics per call -- runs on each batch on its own, in
        that order; the GRU and the two linears behind it, which treat every sample on its own, run ONCE on both batches side by
        side.  Same scores; half the recurrence launches of the step, and every parameter behind pre_conv is used once in the
        backward instead of twice (no gradient fan-in adds)."""
        fa, fb = self._pre_conv_cl(poses_a), self._pre_conv_cl(poses_b)
        s = self._score(torch.cat([fa, fb], dim=0))
        return s[:fa.shape[0]], s[fa.shape[0]:]
