#!/usr/bin/env python3
"""Same command line as the reference's run.py (run.py:37-58): `python run.py -c config/<task>.json -p test`.
The driver logic lives in mdie_amd.host; the network it builds is the HIP engine's `models.cdan.CDAN`."""
import argparse

from utils.logger import ExperimentLogger
from utils.parser import create_model, define_dataloader, define_dataset, define_network, parse
from utils.reproducibility import set_seed_and_cudnn


def main(config):
    set_seed_and_cudnn()
    logger = ExperimentLogger(config)
    if logger.run_dir():
        print(f"[LOGGER] Run dir: {logger.run_dir()}")
    phase = config["phase"]
    dataset = define_dataset(config[phase]["dataset"])
    dataloader = define_dataloader(dataset, config[phase]["dataloader"]["args"])
    network = define_network(config["model"]["networks"][0])
    model = create_model(config=config, network=network, dataloader=dataloader, logger=logger)
    if phase == "train":
        model.train()
        logger.generate_plots()
    else:
        model.test()
    logger.close()
    return model


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", type=str, default="config/default.json", help="Path to the JSON configuration file")
    ap.add_argument("-p", "--phase", type=str, choices=["train", "test"], default="train", help="Phase to run (train or test)")
    main(parse(ap.parse_args()))
