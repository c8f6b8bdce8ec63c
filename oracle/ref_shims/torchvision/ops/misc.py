"""Restatement of torchvision 0.14 ``ops.misc.ConvNormActivation``:
Sequential(Conv2d(bias = norm is None, padding=(k-1)//2*dilation), norm, act(inplace=True))."""
import torch


class ConvNormActivation(torch.nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=None, groups=1,
                 norm_layer=torch.nn.BatchNorm2d, activation_layer=torch.nn.ReLU, dilation=1,
                 inplace=True, bias=None):
        if padding is None:
            padding = (kernel_size - 1) // 2 * dilation
        if bias is None:
            bias = norm_layer is None
        layers = [torch.nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding,
                                  dilation=dilation, groups=groups, bias=bias)]
        if norm_layer is not None:
            layers.append(norm_layer(out_channels))
        if activation_layer is not None:
            layers.append(activation_layer(inplace=inplace) if inplace is not None else activation_layer())
        super().__init__(*layers)
        self.out_channels = out_channels
