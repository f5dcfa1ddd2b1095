"""Generates a tiny on-disk dataset in the reference's format (docs/guide/config.rst:26-55 of the reference):
data/<name>/record_XXX/frameXXXXXX.jpg (224x224), preprocessed_data.npz, ground_truth.npz, dataset_config.json."""
import json
import os

import numpy as np


def make_dataset(root, name="tiny_test", n_episodes=3, ep_len=24, n_actions=6, seed=0, multi_view=False):
    from PIL import Image
    rs = np.random.RandomState(seed)
    folder = os.path.join(root, "data", name)
    paths, actions, rewards, starts, states, targets = [], [], [], [], [], []
    for e in range(n_episodes):
        rec = os.path.join(folder, "record_%03d" % e)
        os.makedirs(rec, exist_ok=True)
        targets.append(rs.rand(3))
        pos = rs.rand(3)
        for t in range(ep_len):
            # smooth blobs so JPEG round-trips are stable; content depends on the "robot position"
            yy, xx = np.mgrid[0:224, 0:224]
            img = np.stack([127 + 100 * np.sin(xx / (20.0 + 30 * pos[c]) + yy / (25.0 + 10 * c) + t * 0.1) for c in range(3)], -1)
            names = ["frame%06d_%d.jpg" % (t, v + 1) for v in range(2)] if multi_view else ["frame%06d.jpg" % t]
            for v, nm in enumerate(names):
                Image.fromarray(np.clip(img + 10 * v, 0, 255).astype(np.uint8)).save(os.path.join(rec, nm), quality=95)
            paths.append("%s/record_%03d/frame%06d" % (name, e, t))
            actions.append(rs.randint(0, n_actions))
            rewards.append(float(rs.rand() > 0.8))
            starts.append(t == 0)
            states.append(pos.copy())
            pos = np.clip(pos + 0.05 * rs.randn(3), 0, 1)
    np.savez(os.path.join(folder, "preprocessed_data.npz"), actions=np.array(actions), rewards=np.array(rewards),
             episode_starts=np.array(starts))
    np.savez(os.path.join(folder, "ground_truth.npz"), images_path=np.array(paths), ground_truth_states=np.array(states),
             target_positions=np.array(targets))
    with open(os.path.join(folder, "dataset_config.json"), "w") as f:
        json.dump({"relative_pos": False}, f)
    return name, np.array(paths), np.array(actions), np.array(rewards), np.array(starts)
