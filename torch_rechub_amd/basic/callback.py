"""EarlyStopper (API mirror of torch_rechub/basic/callback.py:4-33): the trainer's validation-AUC patience rule.

``stop_training(val_auc, weights)`` returns False while training should go on: an AUC above the best seen resets the
counter and snapshots ``weights`` (deep copy, restored by the trainer when it stops, ctr_trainer.py:135); otherwise the
counter advances and the call returns True once ``patience`` evaluations in a row brought no improvement.
"""
from copy import deepcopy


class EarlyStopper(object):

    def __init__(self, patience):
        self.patience = patience
        self.trial_counter, self.best_auc, self.best_weights = 0, 0, None

    def stop_training(self, val_auc, weights):
        improved = val_auc > self.best_auc
        if improved:
            self.best_auc, self.trial_counter = val_auc, 0
            self.best_weights = deepcopy(weights)
        elif self.trial_counter + 1 < self.patience:
            self.trial_counter += 1
        else:
            return True
        return False
