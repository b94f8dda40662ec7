"""env.get_dataset(quality) for the single-env classes: the reference's behaviour-policy
episode loops (chemical_reactor.py:324-420, power_grid.py:194-249, robot_assembly.py:246-308)
run over this package's env.step.  Policy draws come from the global np.random stream in the
reference's call order, so a seeded run follows the reference's own sequence of draws.

Known deviation (SURVEY 8c "G5"): upstream hands float64 actions to step(), which makes part
of its arithmetic float64; the device path pins float32 actions, so per-step values agree to
~1e-7 relative, not bit for bit, and a long trajectory can part ways at a threshold.
The batched, device-resident generator is BatchedIndustrialEnv.rollout with an observation
trajectory (D4RL row-major layout).
"""
import numpy as np


def _cr(env, quality):
    if quality == "expert":
        n_episodes, n_steps, noise_level = 100, 400, 0.1
    elif quality == "medium":
        n_episodes, n_steps, noise_level = 200, 350, 0.3
    elif quality == "mixed":
        n_episodes, n_steps, noise_level = 300, 300, 0.5
    else:
        n_episodes, n_steps, noise_level = 500, 200, 1.0
    observations, actions, rewards, terminals = [], [], [], []
    for _ in range(n_episodes):
        obs, _ = env.reset()
        ep_obs, ep_actions, ep_rewards, ep_terminals = [obs], [], [], []
        for _ in range(n_steps):
            if quality == "expert":
                temp_error = (obs[0] - env.temp_target) / 50
                level_error = (obs[10] - 55) / 50
                action = np.array([
                    -temp_error * 0.5 + np.random.normal(0, noise_level * 0.1),
                    temp_error * 0.3 + np.random.normal(0, noise_level * 0.1),
                    -level_error * 0.2 + np.random.normal(0, noise_level * 0.1)])
            else:
                if np.random.random() < (1 - noise_level):
                    temp_error = (obs[0] - env.temp_target) / 50
                    action = np.array([
                        -temp_error * 0.2 + np.random.normal(0, noise_level * 0.3),
                        np.random.normal(0, noise_level * 0.5),
                        np.random.normal(0, noise_level * 0.3)])
                else:
                    action = np.random.uniform(-1, 1, 3)
            action = np.clip(action, -1, 1)
            next_obs, reward, terminated, truncated, _ = env.step(action)
            done = terminated or truncated
            ep_actions.append(action); ep_rewards.append(reward); ep_terminals.append(done)
            if not done:
                ep_obs.append(next_obs); obs = next_obs
            else:
                break
        n = min(len(ep_actions), len(ep_rewards), len(ep_terminals))
        observations.extend(ep_obs[:n]); actions.extend(ep_actions[:n])
        rewards.extend(ep_rewards[:n]); terminals.extend(ep_terminals[:n])
    terminals = np.array(terminals, dtype=bool)
    return {"observations": np.array(observations, dtype=np.float32), "actions": np.array(actions, dtype=np.float32),
            "rewards": np.array(rewards, dtype=np.float32), "terminals": terminals,
            "timeouts": np.zeros_like(terminals, dtype=bool)}


def _episodic(env, n_samples, policy, clip=None):
    observations, actions, rewards, terminals = [], [], [], []
    for _ in range(n_samples // 1000):
        obs, _ = env.reset()
        done, episode_length = False, 0
        while not done and episode_length < 1000:
            action = policy(obs)
            if clip is not None:
                action = np.clip(action, -clip, clip)
            observations.append(obs.copy()); actions.append(action)
            obs, reward, terminated, truncated, _ = env.step(action)
            rewards.append(reward); terminals.append(terminated)
            done = terminated or truncated
            episode_length += 1
    return {"observations": np.array(observations, dtype=np.float32), "actions": np.array(actions, dtype=np.float32),
            "rewards": np.array(rewards, dtype=np.float32), "terminals": np.array(terminals, dtype=bool)}


def _pg(env, quality):
    n_samples = {"expert": 100000, "medium": 150000, "mixed": 200000, "random": 80000}[quality]
    A = env.action_dim

    def policy(obs):
        if quality == "expert":
            freq_error = obs[0]
            imbalance = np.sum(obs[17:25]) - np.sum(obs[9:17])
            return -0.5 * freq_error * np.ones(A) + 0.1 * imbalance / A
        if quality == "random":
            return np.random.uniform(-5, 5, A)
        if np.random.rand() < 0.6:
            return -0.3 * obs[0] * np.ones(A)
        return np.random.uniform(-3, 3, A)
    return _episodic(env, n_samples, policy)


def _ra(env, quality):
    n_samples = {"expert": 120000, "medium": 180000, "mixed": 250000, "random": 100000}[quality]
    A = env.action_dim

    def policy(obs):
        if quality == "expert":
            error = env.target_position - obs[0:3]
            action = np.concatenate([2.0 * error[:3], -0.1 * obs[7:14][3:]])
            return action[:7]
        if quality == "random":
            return np.random.uniform(-1, 1, A)
        if np.random.rand() < 0.7:
            error = env.target_position - obs[0:3]
            return np.concatenate([1.0 * error[:3], np.random.uniform(-0.5, 0.5, 4)])
        return np.random.uniform(-0.8, 0.8, A)
    return _episodic(env, n_samples, policy, clip=2.0)


def get_dataset(env, quality="mixed"):
    kind = type(env).ENV_ID
    if quality not in ("expert", "medium", "mixed", "random"):
        if kind == "ChemicalReactor-v0":
            quality = quality      # upstream treats any other string as 'random' (chemical_reactor.py:345-347)
        else:
            raise KeyError(quality)  # upstream dict lookup
    if kind == "ChemicalReactor-v0":
        return _cr(env, quality)
    if kind == "PowerGrid-v0":
        return _pg(env, quality)
    return _ra(env, quality)
