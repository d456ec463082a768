"""EarlyStopper (API mirror of torch_rechub/basic/callback.py:4-33)."""
import copy


class EarlyStopper(object):
    """Stop when validation AUC has not improved for ``patience`` evaluations; keeps the best weights."""

    def __init__(self, patience):
        self.patience = patience
        self.trial_counter = 0
        self.best_auc = 0
        self.best_weights = None

    def stop_training(self, val_auc, weights):
        if val_auc > self.best_auc:
            self.best_auc = val_auc
            self.trial_counter = 0
            self.best_weights = copy.deepcopy(weights)
            return False
        if self.trial_counter + 1 < self.patience:
            self.trial_counter += 1
            return False
        return True
