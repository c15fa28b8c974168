/* Shared declarations of the C oracle (oracle/seir_oracle.c: density; oracle/mcmc_oracle.c: sampler).
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/. */
#ifndef SEIR_ORACLE_H
#define SEIR_ORACLE_H

typedef struct {
    int M, T;
    const double *Cstar;      /* [M*M] row-major */
    const double *N;          /* [M] */
    const double *W;          /* [T] */
    const double *weekday_c;  /* [T] */
    const double *log_area_c; /* [M] */
    const double *car_Q;      /* [M*M] */
    double car_half_logdet;
    const double *init_state; /* [M*4] */
} oracle_consts;

/* value (grad == NULL) or value + gradient w.r.t. u of the joint log-prob (covid19uk/inference/inference.py:537-557) */
double seir_oracle_eval(const oracle_consts *c, const double *u, const double *events, int stable, double *grad);

#endif
