from torch.utils.data import DataLoader


def build_data_loader(dataset, data_cfg):
    return DataLoader(dataset, batch_size=data_cfg.get("BATCH_SIZE", 1), shuffle=data_cfg.get("SHUFFLE", False), num_workers=0)
