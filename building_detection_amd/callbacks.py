"""Keras Callback protocol + the schedulers / checkpointing of train_model/DeepLabv3plus.py:627-801.

Host logic only (float64 numpy scalars, like the reference).  `K.set_value(self.model.optimizer.lr, lr)` in
the reference becomes an attribute write on `model.optimizer.lr`; the `backend` shim below keeps the
reference's spelling working.
"""
from __future__ import annotations

import os

import numpy as np


class Callback:
    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None): pass
    def on_train_end(self, logs=None): pass
    def on_epoch_begin(self, epoch, logs=None): pass
    def on_epoch_end(self, epoch, logs=None): pass
    def on_batch_begin(self, batch, logs=None): pass
    def on_batch_end(self, batch, logs=None): pass


class backend:
    """The `tensorflow.keras.backend` calls the reference's callbacks make (DeepLabv3plus.py:658,670,726,736):
    `K.epsilon()`, `K.get_value(model.optimizer.lr)`, `K.set_value(model.optimizer.lr, lr)`."""

    @staticmethod
    def epsilon():
        return 1e-7

    @staticmethod
    def get_value(v):
        return float(v)

    @staticmethod
    def set_value(ref, value):
        if not hasattr(ref, "assign"):
            raise TypeError(f"K.set_value: {ref!r} is not an engine variable (expected model.optimizer.lr)")
        ref.assign(value)


def set_lr(model, value):
    model.optimizer.lr = float(value)


def exponential_fall(global_epoch, lr_base=1e-3, decay=0.9, min_lr=0):
    """DeepLabv3plus.py:627-633."""
    return max(lr_base * pow(decay, global_epoch), min_lr)


class ExponentDecayScheduler(Callback):
    """DeepLabv3plus.py:636-675 (instantiated at :678 but not passed to fit in the reference)."""

    def __init__(self, learning_rate, decay, global_epoch_init=0, min_learning_rate=0, verbose=0, lr_flag=True):
        super().__init__()
        self.learning_rate, self.decay = learning_rate, decay
        self.global_epochs = global_epoch_init
        self.min_learning_rate = min_learning_rate
        self.all_lr_num = []
        self.verbose, self.print_lr = verbose, lr_flag

    def on_epoch_end(self, epoch, logs=None):
        self.global_epochs += 1
        self.all_lr_num.append(float(self.model.optimizer.lr))
        if self.print_lr:
            print('\n EPOCH:%d => learning rate now %6f ' % (self.global_epochs, float(self.model.optimizer.lr)))

    def on_epoch_begin(self, epoch, logs=None):
        set_lr(self.model, exponential_fall(self.global_epochs, self.learning_rate, self.decay, self.min_learning_rate))


def cosine_decay_with_warmup(global_step, learning_rate_base, total_steps, warmup_learning_rate=0.0,
                             warmup_steps=0, min_learn_rate=0):
    """DeepLabv3plus.py:683-702: linear warm-up then half-cosine to zero, float64 numpy on the host."""
    if (global_step > warmup_steps) | (global_step == warmup_steps):
        learning_rate = 0.5 * learning_rate_base * (
            1 + np.cos(np.pi * (global_step - warmup_steps) / float(total_steps - warmup_steps)))
        return max(learning_rate, min_learn_rate)
    k = (learning_rate_base - warmup_learning_rate) / warmup_steps
    return max(k * global_step + warmup_learning_rate, min_learn_rate)


class WarmUpCosineDecayScheduler(Callback):
    """DeepLabv3plus.py:705-739: sets optimizer.lr before every batch, counts global steps after it."""

    def __init__(self, learning_rate_base, total_steps, global_step_init=0, warmup_learning_rate=0.0,
                 warmup_steps=0, min_learn_rate=0, verbose=0):
        super().__init__()
        self.learning_rate_base, self.total_steps = learning_rate_base, total_steps
        self.global_step = global_step_init
        self.warmup_learning_rate, self.warmup_steps = warmup_learning_rate, warmup_steps
        self.verbose, self.min_learn_rate = verbose, min_learn_rate
        self.learning_rates = []

    def on_batch_end(self, batch, logs=None):
        self.global_step += 1
        self.learning_rates.append(float(self.model.optimizer.lr))

    def on_batch_begin(self, batch, logs=None):
        lr = cosine_decay_with_warmup(self.global_step, self.learning_rate_base, self.total_steps,
                                      self.warmup_learning_rate, self.warmup_steps, self.min_learn_rate)
        set_lr(self.model, lr)
        if self.verbose > 0:
            print('\nBatch %05d: setting learning rate to %s.' % (self.global_step + 1, lr))


class MY_EarlyStoppingAtMinLoss(Callback):
    """DeepLabv3plus.py:762-798: despite the name it saves the weights after EVERY epoch (the early-stop
    logic is commented out in the reference) and records `val_PA`."""

    def __init__(self, patience=10, directory="weights1"):
        super().__init__()
        self.patience = patience * 2
        self.add_time = 0
        self.stopped_epoch = 0
        self.need_stopping = False
        self.all_acc = []
        self.directory = directory

    def on_train_begin(self, logs=None):
        self.best_acc = 0

    def on_epoch_end(self, epoch, logs=None):
        self.stopped_epoch += 1
        self.all_acc.append((logs or {}).get('val_PA'))
        os.makedirs(self.directory, exist_ok=True)
        self.model.save_weights(os.path.join(self.directory, "epoch_{}_weights.h5".format(self.stopped_epoch)))

    def on_train_end(self, logs=None):
        if self.need_stopping:
            print("Epoch {}:early stopping".format(self.stopped_epoch))
