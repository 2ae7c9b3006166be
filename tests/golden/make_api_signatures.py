#!/usr/bin/env python3
"""API-signature fixture from the reference's AST (no import of the reference needed).

    python tests/golden/make_api_signatures.py      # writes tests/golden/api_signatures.json

For every class / function of the drop-in surface the positional parameter names and the source
text of their defaults, as the reference declares them.  tests/test_api_signatures_cpu.py compares them with
`inspect.signature` of this repo's objects: same names in the same order with the same defaults; extras are allowed only
as additional keyword parameters WITH defaults after the reference's.
"""
import ast
import json
import os

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))

# reference file -> (import path here, names: 'Class', 'Class.method' or 'function')
SURFACE = {
    'histogram_classes/RGBuvHistBlock.py': ('histogram_classes.RGBuvHistBlock', ['RGBuvHistBlock', 'RGBuvHistBlock.forward']),
    'histogram_classes/rgChromaHistBlock.py': ('histogram_classes.rgChromaHistBlock', ['rgChromaHistBlock', 'rgChromaHistBlock.forward']),
    'histogram_classes/LabHistBlock.py': ('histogram_classes.LabHistBlock', ['LabHistBlock', 'LabHistBlock.forward']),
    'histoGAN/histoGAN.py': ('histoGAN.histoGAN', [
        'Trainer', 'Trainer.train', 'Trainer.evaluate', 'Trainer.generate_truncated', 'Trainer.load', 'Trainer.save',
        'Trainer.clear', 'Trainer.set_data_src', 'Trainer.print_log', 'Trainer.init_GAN', 'Trainer.model_name',
        'HistoGAN', 'HistoGAN.EMA', 'HistoGAN.reset_parameter_averaging',
        'Generator', 'Generator.forward', 'GeneratorBlock', 'GeneratorBlock.forward', 'GeneratorBlock.forward_',
        'RGBBlock', 'RGBBlock.forward', 'Conv2DMod', 'Conv2DMod.forward', 'Discriminator', 'Discriminator.forward',
        'DiscriminatorBlock', 'DiscriminatorBlock.forward', 'HistVectorizer', 'HistVectorizer.forward',
        'StyleVectorizer', 'StyleVectorizer.forward', 'gradient_penalty', 'AugWrapper.forward']),
    'ReHistoGAN/rehistoGAN.py': ('ReHistoGAN.rehistoGAN', ['recoloringTrainer', 'recoloringTrainer.train',
                                                            'recoloringTrainer.evaluate', 'recoloringTrainer.load',
                                                            'recoloringTrainer.set_data_src', 'recoloringGAN']),
    'utils/diff_augment.py': ('utils.diff_augment', ['DiffAugment']),
}


def params_of(fn):
    a = fn.args
    names = [x.arg for x in a.posonlyargs + a.args]
    defaults = [None] * (len(names) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
    out = [{'name': n, 'default': d} for n, d in zip(names, defaults)]
    if names and names[0] == 'self':
        out = out[1:]
    return {'params': out, 'varargs': a.vararg.arg if a.vararg else None, 'kwargs': a.kwarg.arg if a.kwarg else None,
            'kwonly': [{'name': k.arg, 'default': ast.unparse(d) if d is not None else None}
                       for k, d in zip(a.kwonlyargs, a.kw_defaults)]}


def main():
    out = {}
    for rel, (mod, names) in SURFACE.items():
        with open(os.path.join(REF, rel)) as f:
            tree = ast.parse(f.read())
        top = {n.name: n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef))}
        for name in names:
            cls, _, meth = name.partition('.')
            node = top[cls]
            if isinstance(node, ast.ClassDef):
                fns = {n.name: n for n in node.body if isinstance(n, ast.FunctionDef)}
                fn = fns[meth or '__init__']
            else:
                fn = node
            out[f'{mod}:{name}'] = dict(params_of(fn), source=f'{rel}:{fn.lineno}')
    with open(os.path.join(HERE, 'api_signatures.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(len(out), 'signatures')


if __name__ == '__main__':
    main()
