/* ngp_oracle.h — CPU oracle (test infrastructure only; see ngp_oracle.c header). */
#ifndef NGP_ORACLE_H
#define NGP_ORACLE_H
#include "../include/ngp.h"
#ifdef __cplusplus
extern "C" {
#endif
void   ngpo_default_spec(ngp_spec *s);
int    ngpo_kernel_check(const ngp_kernel *k);
double ngpo_kernel_eval(const ngp_spec *s, const ngp_kernel *k, double t1, double t2);
int    ngpo_cov(const ngp_spec *s, const ngp_kernel *k, int n1, const double *t1, int n2,
                const double *t2, int add_diag, double *out);
int    ngpo_chol(int n, double *a, int lda);
int    ngpo_logml(const ngp_spec *s, const ngp_kernel *k, int n, const double *t, const double *y,
                  double *logml);
int    ngpo_predict(const ngp_spec *s, const ngp_kernel *k, int n, const double *t,
                    const double *y, int m, const double *t_new, int noise_on_new, double *mu,
                    double *sigma, double *logml);
int    ngpo_nowcast(const ngp_spec *s, const ngp_kernel *k, int n, const double *t,
                    const double *y, int d, const double *t_add, int D, const double *y_add,
                    int m, const double *t_new, int noise_on_new, double *logml_base,
                    double *logml_full, double *mu, double *sigma);
int    ngpo_logml_grad(const ngp_spec *s, const ngp_kernel *k, int n, const double *t,
                       const double *y, double *logml, double *grad);
int    ngpo_weights_normalize(int P, const double *logw, double *w_norm, double *ess,
                              double *log_norm);
#ifdef __cplusplus
}
#endif
#endif
