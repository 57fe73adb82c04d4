"""Internal glue shared with the reference's surface (gan_lab/_int.py): constants, the pickled
config discovery, and the write-guarded learner config copy."""
import argparse
import copy
import pickle
from pathlib import Path

SUPPORTED_LEARNERS = ('GANLearner', 'ProGANLearner', 'StyleGANLearner',)
SUPPORTED_ARCHS = {
    'Generators': ('Generator32PixResnet', 'Generator64PixResnet', 'ProGenerator', 'StyleGenerator',),
    'Discriminators': ('Discriminator32PixResnet', 'Discriminator64PixResnet', 'ProDiscriminator',
                       'StyleDiscriminator',),
    'AC Discriminators': ('DiscriminatorAC32PixResnet', 'DiscriminatorAC64PixResnet',),
}
FMAP_SAMPLES = 3
RES_INIT = 4


class NotConfiguredError(Exception):
    """Raised when either config.py or data_config.py has not been run."""
    pass


def get_current_configuration(cfg, raise_exception=True):
    """~/.configs_dir.txt -> <dir>/.config.p | .data_config.p  (_int.py:55-84)."""
    try:
        with open(str(Path.home() / '.configs_dir.txt'), 'rb') as f:
            configs_dir = f.readline().decode('utf8').strip()
    except FileNotFoundError:
        if raise_exception:
            raise NotConfiguredError('Please run config.py or data_config.py atleast once before running this '
                                     'function.')
        return None
    if cfg == 'config':
        pickled, module = configs_dir + '/.config.p', 'config.py'
    elif cfg == 'data_config':
        pickled, module = configs_dir + '/.data_config.p', 'data_config.py'
    else:
        raise ValueError("Input configuration does not exist. Options are 'config' and 'data_config'.")
    try:
        with open(pickled, 'rb') as f:
            return pickle.load(f)
    except FileNotFoundError:
        if raise_exception:
            raise NotConfiguredError(f'Please run {module} atleast once in order to obtain your desired '
                                     f'configuration.')
        return None


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.casefold() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.casefold() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


class LearnerConfigCopy(object):
    """Deep copy of the argparse Namespace whose structural attributes cannot be re-assigned
    (_int.py:102-147): non-redefinable -> AttributeError; learner-redefinable -> AttributeError
    pointing at the learner property."""

    def __init__(self, config, learner_class: str, nonredefinable_attrs: tuple,
                 redefinable_from_learner_attrs: tuple):
        assert (isinstance(config, argparse.Namespace) and 'model' in config.__dict__)
        object.__setattr__(self, '__dict__', copy.deepcopy(config.__dict__))
        names = {'GANLearner': 'resnetgan', 'ProGANLearner': 'progan', 'StyleGANLearner': 'stylegan'}
        if learner_class not in names:
            raise ValueError(f"Input learner_class argument set equal to {learner_class}. But currently, "
                             f"learner_class can only be\none of: [ '" + "', '".join(SUPPORTED_LEARNERS) + "' ]")
        self.__dict__['model_name'] = names[learner_class]
        self.__dict__['learner_class'] = learner_class
        self.__dict__['_nonredefinable_attrs'] = nonredefinable_attrs
        self.__dict__['_redefinable_from_learner_attrs'] = redefinable_from_learner_attrs

    def __setattr__(self, name, value):
        if name in self._nonredefinable_attrs:
            msg = f"{self.learner_class}().config.{name} attribute cannot be changed once {self.learner_class} is " \
                  f"instantiated.\n"
            if name == 'model':
                msg += f"Instead, please run 'python config.py {value}' on the command-line and then instantiate " \
                       f"a new {self.learner_class}."
            else:
                msg += f"Instead, please run 'python config.py {self.model_name} --{name}={value}' on the " \
                       f"command-line and then instantiate a new {self.learner_class}."
            raise AttributeError(msg)
        elif name in self._redefinable_from_learner_attrs:
            raise AttributeError(
                f"{self.learner_class}().config.{name} attribute cannot be changed.\n Instead, please change "
                f"{self.learner_class}().{name} to implement this change in the {self.learner_class} instance,\n "
                f"while {self.learner_class}().config.{name} will remain equal to its value when the "
                f"{self.learner_class} was first initialized.")
        else:
            super().__setattr__(name, value)

    def __str__(self):
        hidden = ('_nonredefinable_attrs', '_redefinable_from_learner_attrs', 'model_name', 'learner_class',)
        return ''.join(f'  {k}: {v}\n' for k, v in vars(self).items() if k not in hidden)
