"""BASELINE config 5: the reference demo's `spaceship` mode, re-implemented deterministically.

Follows reference demo/src/demos/spaceship.rs statement by statement — `Actor::{create_enemy :54-92, create_player
:94-118, interact :120-146, update :148-196, transform :198-203, overlaps_with :205-211}`, `enemy_count :213-217`,
`Spaceship::{update_actors :239-262, update_composition_and_cleanup_actors :264-296, compose :321-332}`,
`potatoe_path :337-362`, `ship_path :365-417` — with two documented differences:

* the reference draws its randoms from `StdRng::seed_from_u64(43)` (ChaCha12 from the `rand` crate, not reproducible
  here); this uses SplitMix64 seeded with 43, 24 mantissa bits per draw: `lo + (hi - lo) * (u >> 40) * 2^-24` in f32;
* no keyboard (`keyboard_cmd` = 0) and a fixed `elapsed` = 1/60 s, as SURVEY.md §8(d) C5 prescribes; the 1000 x 1000
  game area is scaled by `scale` (2.16 for 3840 x 2160) and centred horizontally: shapes are built at their scaled size
  (a GeomPresTransform cannot scale up, math/transform.rs:160-222), positions are scaled when the transform is made.

The module is written against forma's public API shape and takes the API module as an argument, so the same code drives
the product (`forma_amd.api`) and, in tests, the oracle-backed mirror (tests/ref_api.py): the two must then produce the
same frames, cache hits included.  All arithmetic is f32 (numpy.float32), like the reference.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
GAME_SIZE = f32(1000.0)
DT_NS = 16_666_667                            # 1/60 s as a Duration


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next_u64(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def gen_range(self, lo: float, hi: float) -> np.float32:
        u = f32(self.next_u64() >> 40) * f32(2.0 ** -24)
        return f32(lo) + (f32(hi) - f32(lo)) * u


class Actor:
    __slots__ = ("kind", "acc", "speed", "pos", "angle", "angle_speed", "max_speed", "friction", "layer", "order", "radius", "alive")


def _v(x, y):
    return np.array([x, y], f32)


class Spaceship:
    def __init__(self, api, width=3840, height=2160, seed=43):
        self.api = api
        self.width, self.height = width, height
        self.scale = f32(min(width, height) / 1000.0)
        self.off_x = f32((width - float(self.scale) * 1000.0) * 0.5)
        self.off_y = f32((height - float(self.scale) * 1000.0) * 0.5)
        self.actors = []
        self.time_ns = 0
        self.rng = SplitMix64(seed)

    # ---- shapes (scaled at build time) --------------------------------------------------------------------------------
    def _pt(self, x, y):
        return self.api.Point(float(f32(x) * self.scale), float(f32(y) * self.scale))

    def potatoe_path(self, x, y, radius):                    # :337-362
        P, g = self._pt, lambda: float(self.rng.gen_range(0.07, 1.4))
        b = self.api.PathBuilder()
        b.move_to(P(x + radius, y))
        b.rat_quad_to(P(x + radius, y - radius), P(x, y - radius), g())
        b.rat_quad_to(P(x - radius, y - radius), P(x - radius, y), g())
        b.rat_quad_to(P(x - radius, y + radius), P(x, y + radius), g())
        b.rat_quad_to(P(x + radius, y + radius), P(x + radius, y), g())
        return b.build()

    def ship_path(self):                                     # :365-417
        P = self._pt
        b = self.api.PathBuilder()
        b.move_to(P(0.0, 50.0)); b.line_to(P(40.0, 50.0)); b.line_to(P(40.0, 60.0))
        b.cubic_to(P(47.0, 56.0), P(54.0, 57.0), P(60.0, 60.0))
        b.line_to(P(60.0, 50.0)); b.line_to(P(80.0, 50.0)); b.line_to(P(80.0, 10.0))
        b.cubic_to(P(67.0, -3.0), P(50.0, -13.0), P(30.0, -20.0))
        b.line_to(P(25.0, -51.0)); b.line_to(P(30.0, -52.0)); b.line_to(P(30.0, -70.0)); b.line_to(P(21.0, -74.0))
        b.cubic_to(P(17.0, -90.0), P(9.0, -102.0), P(0.0, -107.0))
        b.cubic_to(P(-9.0, -102.0), P(-17.0, -90.0), P(-21.0, -74.0))
        b.line_to(P(-30.0, -70.0)); b.line_to(P(-30.0, -52.0)); b.line_to(P(-25.0, -51.0)); b.line_to(P(-30.0, -20.0))
        b.cubic_to(P(-50.0, -13.0), P(-67.0, -3.0), P(-80.0, 10.0))
        b.line_to(P(-80.0, 50.0)); b.line_to(P(-60.0, 50.0)); b.line_to(P(-60.0, 60.0))
        b.cubic_to(P(-54.0, 57.0), P(-47.0, 56.0), P(-40.0, 60.0))
        b.line_to(P(-40.0, 50.0)); b.line_to(P(0.0, 50.0))
        return b.build()

    def _solid(self, r, g, b):
        a = self.api
        return a.Props(fill_rule=a.FillRule.NonZero, func=a.Func.Draw(a.Style(fill=a.Fill.Solid(a.Color(float(r), float(g), float(b), 1.0)))))

    # ---- actors ----------------------------------------------------------------------------------------------------------
    def create_enemy(self, composition):                     # :54-92 (draw order of the randoms kept)
        rng = self.rng
        radius = f32(np.power(rng.gen_range(5.0, 10.0), f32(2.0)))
        x = rng.gen_range(0.0, float(GAME_SIZE))
        angle = rng.gen_range(-0.2, 0.2)
        speed = f32(np.power(rng.gen_range(8.0, 16.0), f32(2.0)))
        r, g, b = rng.gen_range(0.1, 0.3), rng.gen_range(0.1, 0.3), rng.gen_range(0.1, 0.3)
        layer = composition.create_layer()
        layer.insert(self.potatoe_path(0.0, 0.0, radius)).set_props(self._solid(r, g, b))
        a = Actor()
        a.kind = "enemy"; a.pos = _v(x, f32(-0.25) * GAME_SIZE); a.friction = f32(0.0); a.max_speed = f32(np.inf)
        a.speed = _v(np.sin(angle), np.cos(angle)) * speed
        a.acc = _v(0.0, 0.0); a.angle = f32(0.0); a.angle_speed = rng.gen_range(-0.8, 0.8)
        a.radius = radius; a.layer = layer; a.order = None; a.alive = True
        return a

    def create_player(self, composition):                    # :94-118
        layer = composition.create_layer()
        layer.insert(self.ship_path()).set_props(self._solid(0.1, 0.1, 0.1))
        a = Actor()
        a.kind = "player"; a.pos = _v(f32(0.5) * GAME_SIZE, f32(0.9) * GAME_SIZE); a.friction = f32(-1.0)
        a.max_speed = f32(10.0) * GAME_SIZE; a.speed = _v(0.0, 0.0); a.acc = _v(0.0, 0.0)
        a.angle = f32(0.0); a.angle_speed = f32(0.0); a.radius = f32(60.0); a.layer = layer; a.order = None; a.alive = True
        return a

    @staticmethod
    def _interact(a, b):                                     # :120-146
        if not a.alive or not b.alive:
            return
        d = a.radius + b.radius
        u = a.pos - b.pos
        if not (f32(u[0] * u[0] + u[1] * u[1]) < d * d):     # overlaps_with :205-211
            return
        r = b.pos - a.pos
        r_pos_u = r / f32(np.sqrt(f32(r[0] * r[0] + r[1] * r[1])))
        u1 = f32(r_pos_u[0] * a.speed[0] + r_pos_u[1] * a.speed[1])
        u2 = f32(r_pos_u[0] * b.speed[0] + r_pos_u[1] * b.speed[1])
        if u2 > u1:
            return
        m1, m2 = f32(np.power(a.radius, f32(3.0))), f32(np.power(b.radius, f32(3.0)))
        v1 = ((m1 - m2) * u1 + f32(2.0) * m2 * u2) / (m1 + m2)
        v2 = ((m2 - m1) * u2 + f32(2.0) * m1 * u1) / (m1 + m2)
        a.speed = a.speed + r_pos_u * (-u1 + v1)
        b.speed = b.speed + r_pos_u * (-u2 + v2)

    @staticmethod
    def _update(a, delta_t):                                 # :148-196, no keys pressed
        if not a.alive:
            return
        if a.kind == "player":
            a.acc = f32(2000.0) * _v(0.0, 0.0)
            a.pos = np.clip(a.pos, f32(0.0), GAME_SIZE).astype(f32)
            a.angle = f32(np.clip(np.arcsin(f32(-2.0) * a.speed[0] / a.max_speed), f32(-0.2), f32(0.2)))
        a.acc = a.acc + a.speed * a.friction
        a.speed = np.clip(a.speed + a.acc * delta_t, -a.max_speed, a.max_speed).astype(f32)
        a.pos = (a.pos + a.speed * delta_t).astype(f32)
        a.angle = f32(a.angle + a.angle_speed * delta_t)
        a.alive = bool(a.alive and a.pos[0] > f32(-0.5) * GAME_SIZE and a.pos[0] < f32(1.5) * GAME_SIZE
                       and a.pos[1] > f32(-0.5) * GAME_SIZE and a.pos[1] < f32(1.5) * GAME_SIZE)

    def _transform(self, a):                                 # :198-203, position scaled into the canvas
        c, s = f32(np.cos(a.angle)), f32(np.sin(a.angle))
        return self.api.GeomPresTransform.try_from([float(c), float(s), float(-s), float(c),
                                                    float(a.pos[0] * self.scale + self.off_x), float(a.pos[1] * self.scale + self.off_y)])

    @staticmethod
    def enemy_count(time_ns: int) -> int:                    # :213-217
        t = f32(time_ns / 1e9)
        return int(f32(0.0005) * t * t + f32(2.0) * t)

    # ---- one frame ------------------------------------------------------------------------------------------------------
    def compose(self, composition, elapsed_ns: int = DT_NS):   # :321-332
        if not self.actors:
            self.actors.append(self.create_player(composition))
        # update_actors :239-262
        new_enemies = self.enemy_count(self.time_ns + elapsed_ns) - self.enemy_count(self.time_ns)
        for _ in range(new_enemies):
            self.actors.append(self.create_enemy(composition))
        for i in range(len(self.actors)):
            for j in range(i + 1, len(self.actors)):
                self._interact(self.actors[i], self.actors[j])
        dt = f32(elapsed_ns / 1e9)
        for a in self.actors:
            self._update(a, dt)
        self.time_ns += elapsed_ns
        # update_composition_and_cleanup_actors :264-296
        Order = self.api.Order
        for order, a in enumerate(self.actors):
            if a.alive:
                if a.order is not None:
                    if a.order != order:
                        layer = composition.remove(Order(a.order))
                        composition.insert(Order(order), layer)
                else:
                    composition.insert(Order(order), a.layer)
                    a.layer = None
                a.order = order
                composition.get_mut(Order(order)).set_transform(self._transform(a))
            elif a.order is not None:
                composition.remove(Order(a.order))
        self.actors = [a for a in self.actors if a.alive]
