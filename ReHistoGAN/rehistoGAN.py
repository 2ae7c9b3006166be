"""Drop-in for the reference's ReHistoGAN/rehistoGAN.py: same import path and public names
(`from ReHistoGAN import recoloringTrainer`, rehistoGAN.py:17 of the reference's CLI).  Implementation:
histogan_amd/{renets,retrainer,reops}.py over the HIP kernels of histogan_amd/csrc/."""
from histogan_amd.nets import Conv2DMod, Discriminator, GeneratorBlock, HistVectorizer
from histogan_amd.renets import DecoderBlock, EncoderBlock, RecoloringEncoderDecoder, RecoloringGAN
from histogan_amd.retrainer import (NanException, gaussian_op, get_gaussian_kernel, laplacian_op, recoloringGAN,
                                    recoloringTrainer, reconstruction_loss, sobel_op)
from histogan_amd.trainer import gradient_penalty

__all__ = ['recoloringTrainer', 'recoloringGAN', 'RecoloringGAN', 'RecoloringEncoderDecoder', 'EncoderBlock',
           'DecoderBlock', 'reconstruction_loss', 'get_gaussian_kernel', 'gaussian_op', 'laplacian_op', 'sobel_op',
           'gradient_penalty', 'NanException', 'GeneratorBlock', 'HistVectorizer', 'Discriminator', 'Conv2DMod']
