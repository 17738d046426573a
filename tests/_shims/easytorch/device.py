import torch


def to_device(x, device=None):
    if device is None:
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    return x.to(device) if torch.is_tensor(x) else x
