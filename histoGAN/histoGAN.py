"""Drop-in for the reference's histoGAN/histoGAN.py: same import path and public names
(`from histoGAN import Trainer, NanException`, histoGAN.py:18 of the reference's CLI; the other names
are imported by ReHistoGAN/rehistoGAN.py:34 and the projection scripts).  Implementation:
histogan_amd/{nets,trainer,optim,ddp}.py over the HIP kernels of histogan_amd/csrc/."""
from histogan_amd.augment import AugWrapper
from histogan_amd.nets import (Conv2DMod, Discriminator, DiscriminatorBlock, Generator, GeneratorBlock,
                               HistVectorizer, RGBBlock, StyleVectorizer)
from histogan_amd.trainer import (EMA, HistoGAN, NanException, Trainer, evaluate_in_chunks, gradient_penalty,
                                  latent_to_w, styles_def_to_tensor)

__all__ = ['Trainer', 'HistoGAN', 'NanException', 'Generator', 'GeneratorBlock', 'HistVectorizer', 'RGBBlock',
           'Conv2DMod', 'Discriminator', 'DiscriminatorBlock', 'StyleVectorizer', 'EMA', 'gradient_penalty',
           'latent_to_w', 'styles_def_to_tensor', 'evaluate_in_chunks', 'AugWrapper']
