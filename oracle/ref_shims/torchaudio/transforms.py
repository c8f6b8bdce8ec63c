"""Restatement of torchaudio 0.13 ``transforms.{Frequency,Time}Masking`` for a 3-D
(B, F, T) input: one mask shared by the batch (iid_masks only applies to 4-D input)."""
import torch


class _AxisMasking(torch.nn.Module):
    def __init__(self, mask_param, axis, iid_masks=False):
        super().__init__()
        self.mask_param, self.axis, self.iid_masks = mask_param, axis, iid_masks

    def forward(self, specgram, mask_value=0.0):
        size = specgram.size(self.axis)
        value = torch.rand(1) * self.mask_param
        min_value = torch.rand(1) * (size - value)
        start = int(min_value.long())
        end = start + int(value.long())
        idx = torch.arange(size, device=specgram.device)
        mask = (idx >= start) & (idx < end)
        if self.axis == 1:
            mask = mask.unsqueeze(-1)
        return specgram.masked_fill(mask, mask_value)


class FrequencyMasking(_AxisMasking):
    def __init__(self, freq_mask_param, iid_masks=False):
        super().__init__(freq_mask_param, 1, iid_masks)


class TimeMasking(_AxisMasking):
    def __init__(self, time_mask_param, iid_masks=False, p=1.0):
        super().__init__(time_mask_param, 2, iid_masks)
