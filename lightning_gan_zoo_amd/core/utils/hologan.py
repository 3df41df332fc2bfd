"""HoloGAN learning-rate schedule, reference core/utils/hologan.py:3-9: constant for the first half of
training, then linear decay to zero."""
from torch.optim.lr_scheduler import LambdaLR


def create_hologan_lr_scheduler(total_epochs, optimizer):
    half = total_epochs / 2

    def factor(epoch):
        return 1 if epoch <= half else 1 - ((epoch - half) / half)

    return LambdaLR(optimizer, factor)
