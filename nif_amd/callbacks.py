"""The Keras callback protocol the reference's README uses (README.md:71-97)."""


class Callback(object):
    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass


class LearningRateScheduler(Callback):
    """tf.keras.callbacks.LearningRateScheduler(schedule): lr = schedule(epoch, lr) at epoch begin."""

    def __init__(self, schedule, verbose=0):
        super().__init__()
        self.schedule = schedule
        self.verbose = verbose

    def on_epoch_begin(self, epoch, logs=None):
        opt = self.model.optimizer
        lr = float(self.schedule(epoch, opt.learning_rate))
        opt.learning_rate = lr
        if self.verbose:
            print("Epoch %d: LearningRateScheduler setting learning rate to %g." % (epoch + 1, lr))

    def on_epoch_end(self, epoch, logs=None):
        if logs is not None:
            logs["lr"] = self.model.optimizer.learning_rate
