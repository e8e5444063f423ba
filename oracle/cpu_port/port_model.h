/* ORACLE / CPU baseline (test infrastructure): model table shared by the generated code and solver_port.c */
#ifndef PORT_MODEL_H
#define PORT_MODEL_H
#define PORT_MAXQ 4
#define PORT_MAXCLS 4
typedef void (*cost_fn)(const double*, const double*, double*, double*, double*);
typedef void (*costT_fn)(const double*, double*, double*, double*);
typedef void (*dyn_fn)(const double*, const double*, const double*, const double*, double*, double*, double*, double*,
                       double*, double*);
typedef void (*dynres_fn)(const double*, const double*, const double*, double*);
typedef void (*costval_fn)(const double*, const double*, double*);
typedef void (*costTval_fn)(const double*, double*);
typedef void (*con_fn)(const double*, const double*, const double*, double*, double*, double*);
typedef void (*conval_fn)(const double*, const double*, double*);
typedef struct {
  int nc, np;            /* rows, variables read ([x;u] or [x]) */
  int ineq[PORT_MAXQ];   /* 1: row is c(x,u) <= 0 */
  con_fn f;
  conval_fn val;
} port_con_class;
typedef struct {
  const char* name;
  int n, m;
  cost_fn cost; costT_fn costT; dyn_fn dyn; dynres_fn dynres; costval_fn costval; costTval_fn costTval;
  int n_class;
  port_con_class cls[PORT_MAXCLS];
} port_model;
extern const port_model PORT_MODELS[];
extern const int PORT_NMODELS;
#endif
